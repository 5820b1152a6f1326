"""More than one device behind the C ABI: rt_multi_* (one process, one host thread and one scene
handle per device, samples dealt to the members) and rt_comm_* / rt_reduce_and_resolve (the RCCL
reduce of the members' sums onto the root, then the resolve).  A one-GPU machine can run RCCL with
one member only; the partition itself is exercised with several members sharing the card and the
sums meeting in host memory (RT_REDUCE_HOST).  Bars: counters exact; sums equal to the one-device
frame up to the order of fp64 additions (1e-12 relative); resolved bytes equal."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import hip, scenes
from raytracer_amd.desc import render_params

pytestmark = pytest.mark.gpu


def _frame():
    desc, cam = scenes.bunny_cornell_scene(nu=40, nv=41, light="face", extras=True)
    return desc, hip.camera_init(**cam), render_params(96, 64, 4, 3, seed=5, max_jump=16)


def test_virtual_members_on_one_card_add_up_to_the_single_device_frame():
    desc, c, prm = _frame()
    one, k1 = hip.HipScene(desc).render(c, prm)
    for members in (2, 3, 5):
        ms = hip.MultiScene(desc, [0] * members, hip.REDUCE_HOST)
        img, bgr, k = ms.render(c, prm)
        assert np.array_equal(k, k1)                                    # every (pixel, sample) rendered exactly once
        assert np.allclose(img, one, rtol=1e-12, atol=1e-14)
        assert np.array_equal(bgr, hip.resolve_bgr8_host(img, 12))
        ms.close()
    # more members than samples: some slices are empty
    ms = hip.MultiScene(desc, [0] * 4, hip.REDUCE_HOST)
    small = render_params(40, 30, 1, 3, seed=5)
    img, _, k = ms.render(c, small)
    ref, kr = hip.HipScene(desc).render(c, small)
    assert np.array_equal(k, kr) and np.allclose(img, ref, rtol=1e-12, atol=1e-14)
    # a sub-range of the table, and a tone operator on the root
    part = render_params(96, 64, 4, 3, sample_begin=2, sample_end=9, seed=5, max_jump=16)
    img, bgr, _ = ms.render(c, part, tone=hip.TONE_SQRT)
    ref, _ = hip.HipScene(desc).render(c, part)
    assert np.allclose(img, ref, rtol=1e-12, atol=1e-14)
    assert np.array_equal(bgr, hip.resolve_bgr8_host(img, 12, hip.TONE_SQRT))


def test_members_share_one_octree_build():
    """rt_multi_create builds the octrees once and hands every member its copy: four members cost about
    what one scene costs (the build dominates), and all four hold the same trees."""
    import time
    desc, cam = scenes.atrium_scene(detail=0.6)
    t0 = time.perf_counter()
    one = hip.HipScene(desc)
    t_one = time.perf_counter() - t0
    t0 = time.perf_counter()
    ms = hip.MultiScene(desc, [0] * 4, hip.REDUCE_HOST)
    t_four = time.perf_counter() - t0
    assert t_four < 2.0 * t_one + 0.5, (t_one, t_four)
    c, prm = hip.camera_init(**cam), render_params(64, 36, 2, 2, seed=9, max_jump=8)
    img, _, k = ms.render(c, prm)
    ref, kr = one.render(c, prm)
    assert np.array_equal(k, kr) and np.allclose(img, ref, rtol=1e-12, atol=1e-14)
    ms.close()


def test_rccl_path_with_one_member():
    """RT_REDUCE_RCCL on the devices this machine has (one): the reduce runs through RCCL."""
    desc, c, prm = _frame()
    one, k1 = hip.HipScene(desc).render(c, prm)
    ms = hip.MultiScene(desc, [0], hip.REDUCE_RCCL)
    for _ in range(2):                                                  # the second frame reuses buffers and communicator
        img, bgr, k = ms.render(c, prm)
        assert np.array_equal(k, k1) and np.allclose(img, one, rtol=1e-12, atol=1e-14)
        assert np.array_equal(bgr, hip.resolve_bgr8_host(img, 12))
    L = hip.lib()
    h = C.c_void_p()
    dup = (C.c_int * 2)(0, 0)
    assert L.rt_comm_create(2, dup, C.byref(h)) != 0 and b"twice" in L.rt_last_error()
    d, keep = desc.to_c()
    assert L.rt_multi_create(C.byref(d), dup, 2, hip.REDUCE_RCCL, C.byref(h)) != 0


def test_reduce_and_resolve_one_process_per_gpu_form():
    """rt_comm_unique_id / rt_comm_init_rank (what a launcher with one process per GPU uses), with
    the one rank this machine allows: sums stay, bytes equal rt_resolve_bgr8's."""
    import torch
    rng = np.random.default_rng(3)
    W, H, spp = 50, 30, 8
    sums = rng.uniform(-0.5, 12.0, (H, W, 3))
    comm = hip.Comm(unique_id=hip.Comm.unique_id(), nranks=1, rank=0, device=0)
    assert comm.size() == 1
    d = torch.from_numpy(sums.copy()).to("cuda:0")
    out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda:0")
    comm.reduce_and_resolve(0, d.data_ptr(), W, H, spp, out.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), sums)
    assert np.array_equal(out.cpu().numpy(), hip.resolve_bgr8_host(sums, spp))
    with pytest.raises(hip.RtError):
        comm.reduce_and_resolve(3, d.data_ptr(), W, H, spp, out.data_ptr())
    comm.close()


def test_bench_starts_its_own_rank_processes_for_more_than_one_gpu():
    """`python bench.py --gpus 2` with no launcher: the script starts one process per rank itself
    and rank 0's line comes out as its own.  Rehearsed here with both ranks on the one card (gloo
    carries the reduce; on a multi-GPU node the same path runs over RCCL, one device per rank)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RT_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--size", "96", "--spp-root", "4", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["unit"] == "Mrays/s" and d["value"] > 0
    assert d["config"]["spp_total"] == 32 and d["config"]["spp_per_gpu"] == 16
    assert len(d["rank_ms_per_step"]) == 2 and max(d["rank_ms_per_step"]) == pytest.approx(d["ms_per_step"])
    assert d["launcher"] == "bench.py's own rank processes" and d["rccl_nranks"] == 0 and "gloo" in d["reduce"]
    assert d["cpu_baseline"] is None and 0 < d["roofline"]["frac"] <= 1
    # more ranks than devices over RCCL is refused with a clear message, before any rendering
    env2 = dict(env); env2.pop("RT_BENCH_BACKEND")
    import torch
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(torch.cuda.device_count() + 1), "--steps", "1", "--warmup", "0",
                          "--size", "96", "--spp-root", "4", "--no-cpu-baseline"], env=env2, capture_output=True, text=True, timeout=280)
    assert bad.returncode != 0 and "devices" in bad.stderr
