"""Counters of the closest-hit engine's launches alone (bounce and shadow rays), for the statistics
variants of the library (-DRT_STEP_STATS=N give the four work counters other meanings; the sweep
kernels' own counts would pollute the totals).  Development aid; run on the GPU box.
    RT_HIP_LIB=raytracer_amd/_var/st6.so python tools/engine_stats.py bunny 512 4"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import hip, scenes  # noqa: E402
from raytracer_amd.desc import render_params  # noqa: E402

name, size, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
desc, cam = scenes.bunny_cornell_scene() if name == "bunny" else scenes.atrium_scene()
W, H = (size, size) if name == "bunny" else (size * 16 // 9, size)
sc = hip.HipScene(desc)
c = hip.camera_init(**cam)
prm = render_params(W, H, n, seed=1)
sc.render(c, prm)
sc.set_profiling(True)
sc.get_kernel_stats(reset=True)
sc.render(c, prm)
for st, v in sc.get_kernel_stats().items():
    if v["launches"]:
        print(name, st, {k: (round(x, 1) if k == "ms" else int(x)) for k, x in v.items()})
