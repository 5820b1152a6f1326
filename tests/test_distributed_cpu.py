"""The N > 1 path on CPU: two gloo ranks each render their slice of the sample indices (the
oracle stands in for the GPU renderer here), the partial sums are reduced to rank 0, and the
result must equal the single-rank frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd.parallel import sample_slice

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sample_slices_partition_exactly():
    for total in (0, 1, 7, 16, 256, 1024, 1000):
        for world in (1, 2, 3, 4, 8):
            sl = [sample_slice(r, world, total) for r in range(world)]
            assert sl[0][0] == 0 and sl[-1][1] == total
            assert all(sl[i][1] == sl[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in sl]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sample_slice(2, 2, 10)


def _worker(rank, world, port, W, H, n, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc
    from raytracer_amd import scenes
    from raytracer_amd.desc import render_params
    from raytracer_amd.parallel import render_sharded
    desc, cam = scenes.three_sphere_scene("face")
    s = orc.OracleScene(desc)
    c = orc.camera_init(**cam)
    buf = torch.zeros(H, W, 3, dtype=torch.float64)

    def render_slice(b, e, t):
        img, _ = s.render(c, render_params(W, H, n, sample_begin=b, sample_end=e, seed=11), threads=2)
        t += torch.from_numpy(img)

    render_sharded(render_slice, n * n, buf)
    if rank == 0:
        np.save(out_path, buf.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_partition_matches_single_rank(tmp_path):
    from oracle import orc
    from raytracer_amd import scenes
    from raytracer_amd.desc import render_params
    W, H, n = 20, 15, 4
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, port, W, H, n, out), nprocs=2, join=True)
    got = np.load(out)
    desc, cam = scenes.three_sphere_scene("face")
    s = orc.OracleScene(desc)
    whole, _ = s.render(orc.camera_init(**cam), render_params(W, H, n, seed=11), threads=2)
    assert np.allclose(got, whole, rtol=1e-13, atol=1e-15)
    # resolve after the reduce: identical bytes
    assert np.array_equal(orc.resolve_bgr8(got, n * n), orc.resolve_bgr8(whole, n * n))


def test_frame_slices_deal_frames_round_robin():
    from raytracer_amd.parallel import frame_slice, frame_times
    for nframes in (0, 1, 7, 8, 134):
        for world in (1, 2, 3, 8):
            sl = [frame_slice(r, world, nframes) for r in range(world)]
            assert sorted(f for s in sl for f in s) == list(range(nframes))
            assert all(f % world == r for r, s in enumerate(sl) for f in s)
    t = frame_times(134)
    acc = 0.0
    for f in range(134):                                   # the running sum of main.cpp:83, not f * 0.015
        assert t[f] == acc
        acc += 0.015
    assert t[133] <= 2.0 < acc


def _animation(scenes_mod):
    """A scene with a MovableFaceLight (light 1) seen through a ChangeFovCamera."""
    desc, cam = scenes_mod.phong_scene("face")
    return desc, cam, 1


def _frame_worker(rank, world, port, W, H, n, nframes, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc
    from raytracer_amd import scenes
    from raytracer_amd.desc import render_params
    from raytracer_amd.parallel import render_frames
    desc, cam, li = _animation(scenes)
    c0 = orc.camera_init(**cam)
    l0 = desc.lights[li].copy()

    def render_frame(f, time):
        desc.lights[li] = orc.light_move_face(l0, time)
        img, _ = orc.OracleScene(desc).render(orc.camera_change_fov(c0, time), render_params(W, H, n, seed=3, max_jump=6), threads=2)
        return orc.resolve_bgr8(img, n * n)

    frames = render_frames(render_frame, nframes)
    if rank == 0:
        np.save(out_path, np.stack(frames))
    else:
        assert frames is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_frame_partition_matches_single_rank(tmp_path):
    """Animation frames dealt round-robin over two gloo ranks (frame f on rank f mod 2), gathered
    in order on rank 0: byte-identical to rendering the frames one after another."""
    from oracle import orc
    from raytracer_amd import scenes
    from raytracer_amd.desc import render_params
    from raytracer_amd.parallel import render_frames
    W, H, n, nframes = 16, 12, 2, 5
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "frames.npy")
    mp.spawn(_frame_worker, args=(2, port, W, H, n, nframes, out), nprocs=2, join=True)
    got = np.load(out)
    desc, cam, li = _animation(scenes)
    c0 = orc.camera_init(**cam)
    l0 = desc.lights[li].copy()

    def render_frame(f, time):
        desc.lights[li] = orc.light_move_face(l0, time)
        img, _ = orc.OracleScene(desc).render(orc.camera_change_fov(c0, time), render_params(W, H, n, seed=3, max_jump=6), threads=2)
        return orc.resolve_bgr8(img, n * n)

    whole = np.stack(render_frames(render_frame, nframes, rank=0, world=1))
    assert got.shape == (nframes, H, W, 3) and np.array_equal(got, whole)
    assert not np.array_equal(got[0], got[3])               # the animation does move
