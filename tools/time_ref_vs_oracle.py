"""The compiled reference's own Octree::hit (oracle/_ref, built from /root/reference) timed beside the
CPU oracle's restatement on the same triangle soup and rays, one thread each (development aid; runs
in the build container only -- the reference does not travel).  Prints one JSON line for profiles/."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.desc import LIGHT_DT, MAT_NORMAL, MATERIAL_DT, OBJECT_DT, RAY_DT, SceneDesc  # noqa: E402

assert orc.ref_available(), "build oracle/_ref first (make -C oracle)"
ntri, nray = 60000, 200000
tris = scenes.soup(ntri, seed=11, edge_lo=3e-3, edge_hi=4e-2)
rng = np.random.default_rng(12)
rays = np.zeros(nray, RAY_DT)
rays["e"] = rng.uniform(-0.5, 1.5, (nray, 3))
d = rng.uniform(0, 1, (nray, 3)) - rays["e"]
rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
tree = orc.RefTree(tris, pad=1e-7, max_size=128)
t0 = time.perf_counter(); prim_r, t_r, tests_r = tree.hit(rays, 0.0, 1e10); ref_s = time.perf_counter() - t0
desc = SceneDesc(scenes._stack([scenes.mesh_object(0, 0)], OBJECT_DT), scenes._stack([scenes.material(MAT_NORMAL, color=(.5, .5, .5))], MATERIAL_DT),
                 np.zeros(0, LIGHT_DT), [tris])
so = orc.OracleScene(desc)
t0 = time.perf_counter(); prim_o, t_o, _, cnt = so.trace_rays(rays, 0.0, 1e10); orc_s = time.perf_counter() - t0
same = bool(np.array_equal(prim_r, prim_o) and np.array_equal(t_r, t_o))
print(json.dumps({"workload": f"Octree::hit over a {ntri}-triangle soup (leaf capacity 128), {nray} rays, one thread, build container (8 cores, g++ -O2)",
                  "reference_seconds": ref_s, "reference_Mrays_per_s": nray / ref_s / 1e6,
                  "oracle_seconds": orc_s, "oracle_Mrays_per_s": nray / orc_s / 1e6,
                  "oracle_over_reference": ref_s / orc_s, "ids_and_distances_identical": same,
                  "primitive_tests": int(np.sum(tests_r))}))
