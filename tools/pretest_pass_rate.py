"""Share of pre-tested records that reach the fp64 primitive test, on the bunny and atrium scenes.
Needs a library built with -DRT_STEP_STATS=5 (tools/build_variant.sh st5 -DRT_STEP_STATS=5, then
RT_HIP_LIB=raytracer_amd/_var/st5.so): counter 0 then counts those records, counter 3 the pre-tests."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from raytracer_amd import hip, scenes
from raytracer_amd.desc import render_params
for name, (desc, cam), (W, H) in (("bunny", scenes.bunny_cornell_scene(), (512, 512)), ("atrium", scenes.atrium_scene(), (480, 270))):
    sg = hip.HipScene(desc)
    c = hip.camera_init(**cam)
    img, k = sg.render(c, render_params(W, H, 4, seed=1243))
    print(name, "counters", k, "cand/pretested = %.4f" % (k[0] / max(k[3], 1)), flush=True)
