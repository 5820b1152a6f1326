"""rt_ray_color_noalloc, the entry point that allocates nothing: its pending-ray stacks live in the
library's static device array.  Called through the C ABI with device buffers and checked against
rt_ray_color on the same rays, twice in a row, the second call with a smaller input (the workspace
keeps the first call's contents).  Bars as for rt_ray_color itself: ray counters exact; radiance
within 1e-12 relative (each ray is its own output element, the same arithmetic on both sides; only
the order of the fp64 atomic additions into it may differ)."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import hip, scenes
from raytracer_amd.desc import RAY_DT, render_params

pytestmark = pytest.mark.gpu


def _camera_rays(desc_cam, W, H, jitter_seed):
    """Rays through a W x H grid of film points (host arithmetic only)."""
    c = hip.camera_init(**desc_cam)
    rng = np.random.default_rng(jitter_seed)
    s = (np.arange(W)[None, :] + rng.uniform(0, 1, (H, W))) / W
    t = (np.arange(H)[:, None] + rng.uniform(0, 1, (H, W))) / H
    e = np.array(c.e); llc = np.array(c.llc); u = np.array(c.u); v = np.array(c.v)
    p = llc + u * (s[..., None] * c.width) + v * (t[..., None] * c.height)
    d = p - e
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    rays = np.zeros(W * H, RAY_DT)
    rays["e"] = e
    rays["d"] = d.reshape(-1, 3)
    return rays


def _run(entry, scene, rays, prm, t0, t1, jumptime):
    import torch
    dev = torch.device("cuda:0")
    n = len(rays)
    d_rays = torch.from_numpy(rays.view(np.float64).reshape(n, 6).copy()).to(dev)
    d_out = torch.full((n, 3), 7.0, dtype=torch.float64, device=dev)        # the call must overwrite, not add to, stale values
    d_cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    rc = entry(scene._h, d_rays.data_ptr(), n, t0, t1, jumptime, C.byref(prm), d_out.data_ptr(), d_cnt.data_ptr(), None)
    assert rc == 0, hip.lib().rt_last_error()
    scene.check()
    return d_out.cpu().numpy(), d_cnt.cpu().numpy()


def test_noalloc_entry_matches_allocating_entry_twice_in_a_row():
    L = hip.lib()
    hip.set_device(0)
    desc, cam = scenes.bunny_cornell_scene(nu=60, nv=61, extras=True)      # octree + mirror + glass: all ray kinds, two pushes and more per hit
    sc = hip.HipScene(desc)
    prm = render_params(1, 1, 1, seed=99, max_jump=24)
    big = _camera_rays(cam, 320, 200, 1)          # 64 000 rays: several rounds through the fixed-size stacks
    small = _camera_rays(cam, 37, 23, 2)          # 851 rays, not a multiple of anything
    for rays, (t0, t1, jt) in ((big, (0.0, 1e10, 0)), (small, (1e-9, 400.0, 3))):
        ref, kref = _run(L.rt_ray_color, sc, rays, prm, t0, t1, jt)
        got, kgot = _run(L.rt_ray_color_noalloc, sc, rays, prm, t0, t1, jt)
        assert np.array_equal(kref, kgot)
        assert int(kgot[0]) > 3 * len(rays)
        err = np.abs(got - ref) / np.maximum(1e-30, np.maximum(np.abs(got), np.abs(ref)))
        assert err.max() <= 1e-12, err.max()
        assert ref.mean() > 0.01
    # another scene through the same workspace right after, and the limits
    desc2, cam2 = scenes.three_sphere_scene("face")
    sc2 = hip.HipScene(desc2)
    rays = _camera_rays(cam2, 64, 48, 3)
    ref, kref = _run(L.rt_ray_color, sc2, rays, prm, 0.0, 1e10, 0)
    got, kgot = _run(L.rt_ray_color_noalloc, sc2, rays, prm, 0.0, 1e10, 0)
    assert np.array_equal(kref, kgot) and np.allclose(got, ref, rtol=1e-12, atol=0)
    import torch
    buf = torch.zeros(8, dtype=torch.float64, device="cuda:0")
    assert L.rt_ray_color_noalloc(sc2._h, buf.data_ptr(), (1 << 20) + 1, 0.0, 1e10, 0, C.byref(prm), buf.data_ptr(), None, None) != 0
    assert b"1048576" in L.rt_last_error()
    assert L.rt_ray_color_noalloc(sc2._h, buf.data_ptr(), 0, 0.0, 1e10, 0, C.byref(prm), buf.data_ptr(), None, None) == 0


def test_noalloc_allocates_nothing_and_two_calls_on_two_streams_in_a_row():
    """On a scene handle that has never rendered, the entry point leaves the handle without a
    workspace and the device's free memory where it was; and a second call on another stream right
    behind the first (no synchronisation by the caller in between: each call returns with its work
    complete) does not disturb the first one's result."""
    import torch
    L = hip.lib()
    hip.set_device(0)
    dev = torch.device("cuda:0")
    desc, cam = scenes.bunny_cornell_scene(nu=60, nv=61, extras=True)
    prm = render_params(1, 1, 1, seed=7, max_jump=24)
    ra, rb = _camera_rays(cam, 200, 120, 11), _camera_rays(cam, 90, 70, 12)
    # kernels loaded and reference results made with another handle first
    other = hip.HipScene(desc)
    ref_a, k_a = _run(L.rt_ray_color, other, ra, prm, 0.0, 1e10, 0)
    ref_b, k_b = _run(L.rt_ray_color, other, rb, prm, 0.0, 1e10, 0)
    _run(L.rt_ray_color_noalloc, other, rb, prm, 0.0, 1e10, 0)
    # the two streams have run the entry point's kernels once (a stream's first use takes queue memory of its own)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    n = len(rb)
    w_r = torch.from_numpy(rb.view(np.float64).reshape(n, 6).copy()).to(dev)
    w_o = torch.zeros(n, 3, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for st in (s1, s2):
        assert L.rt_ray_color_noalloc(other._h, w_r.data_ptr(), n, 0.0, 1e10, 0, C.byref(prm), w_o.data_ptr(), None, st.cuda_stream) == 0
    torch.cuda.synchronize()
    other.close()

    sc = hip.HipScene(desc)
    bufs = []
    for rays in (ra, rb):
        n = len(rays)
        bufs.append((torch.from_numpy(rays.view(np.float64).reshape(n, 6).copy()).to(dev),
                     torch.full((n, 3), 7.0, dtype=torch.float64, device=dev), torch.zeros(4, dtype=torch.int64, device=dev)))
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(dev)[0]
    for (r, o, c), st in zip(bufs, (s1, s2)):
        rc = L.rt_ray_color_noalloc(sc._h, r.data_ptr(), r.shape[0], 0.0, 1e10, 0, C.byref(prm), o.data_ptr(), c.data_ptr(), st.cuda_stream)
        assert rc == 0, L.rt_last_error()
    torch.cuda.synchronize()
    free_after = torch.cuda.mem_get_info(dev)[0]
    assert sc.workspace_bytes() == 0                       # the handle never got a workspace of its own
    assert abs(free_before - free_after) < (8 << 20)       # rt_ray_color would have taken >= 100 MB for its stacks
    for (r, o, c), ref, kref in zip(bufs, (ref_a, ref_b), (k_a, k_b)):
        got = o.cpu().numpy()
        assert np.array_equal(c.cpu().numpy(), kref)
        err = np.abs(got - ref) / np.maximum(1e-30, np.maximum(np.abs(got), np.abs(ref)))
        assert err.max() <= 1e-12, err.max()
