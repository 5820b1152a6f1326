"""Times the primary-ray id buffer of the 1 M-triangle soup at 3840 x 2160 (development aid; run on
the GPU box) and prints one JSON line for profiles/."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from raytracer_amd import hip, scenes  # noqa: E402

desc, cam = scenes.soup_scene(1_000_000)
W, H = 3840, 2160
sc = hip.HipScene(desc)
c = hip.camera_init(**cam)
dev = torch.device("cuda:0")
prim = torch.empty(H * W, dtype=torch.int32, device=dev)
tt = torch.empty(H * W, dtype=torch.float64, device=dev)
cnt = torch.zeros(4, dtype=torch.int64, device=dev)
times = []
for r in range(6):
    cnt.zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sc.trace_primary_dev(c, W, H, prim.data_ptr(), tt.data_ptr(), cnt.data_ptr())
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
k = cnt.cpu().numpy()
best = min(times[1:])
print(json.dumps({"workload": "primary-ray id buffer, 1 000 000-triangle soup (seed 42), 3840x2160, one ray per pixel centre",
                  "ms": best * 1e3, "Mrays_per_s": W * H / best / 1e6, "rays": int(k[0]), "inner_visits": int(k[1]), "leaf_visits": int(k[2]),
                  "primitive_tests": int(k[3]), "hit_pixels": int((prim != -1).sum().item()), "octree": sc.octree_stats(0),
                  "note": "best of 5 after a warm-up launch; ids and distances are checked bit for bit in tests/test_gpu_parity.py"}))
