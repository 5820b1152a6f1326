"""The drop-in boundary without a GPU: the C-ABI library loads, exports every symbol that
include/rt_hip.h declares, its structs have the layout the Python side assumes, host-only
entry points work, and device entry points fail loudly instead of falling back to the CPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import orc
from raytracer_amd import desc as D
from raytracer_amd import hip, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 20
    L = hip.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in rt_hip.h but not exported"
    assert set(hip.EXPORTS) == set(names)


def test_struct_layout_matches_header(tmp_path):
    """sizeof / offsetof straight from the C header vs the numpy / ctypes mirrors."""
    prog = tmp_path / "layout.c"
    prog.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "rt_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(rt_object), sizeof(rt_material), sizeof(rt_light), sizeof(rt_mesh),
         sizeof(rt_scene_desc), sizeof(rt_camera), sizeof(rt_render_params), sizeof(rt_ray));
  printf("%zu %zu %zu %zu %zu %zu\n", offsetof(rt_object, p), offsetof(rt_material, blur), offsetof(rt_material, nr),
         offsetof(rt_light, right), offsetof(rt_light, diffuse), offsetof(rt_render_params, seed));
  printf("%zu %zu %zu\n", offsetof(rt_scene_desc, background), offsetof(rt_scene_desc, tri_box_pad), offsetof(rt_scene_desc, octree_max_depth));
  return 0; }''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    sizes = [int(x) for x in out[0].split()]
    assert sizes == [D.OBJECT_DT.itemsize, D.MATERIAL_DT.itemsize, D.LIGHT_DT.itemsize, C.sizeof(D.MeshC),
                     C.sizeof(D.SceneDescC), C.sizeof(D.CameraC), C.sizeof(D.RenderParamsC), D.RAY_DT.itemsize]
    offs = [int(x) for x in out[1].split()]
    assert offs == [D.OBJECT_DT.fields["p"][1], D.MATERIAL_DT.fields["blur"][1], D.MATERIAL_DT.fields["nr"][1],
                    D.LIGHT_DT.fields["right"][1], D.LIGHT_DT.fields["diffuse"][1], D.RenderParamsC.seed.offset]
    offs = [int(x) for x in out[2].split()]
    assert offs == [D.SceneDescC.background.offset, D.SceneDescC.tri_box_pad.offset, D.SceneDescC.octree_max_depth.offset]


def test_camera_init_is_host_arithmetic_and_matches_oracle():
    rng = np.random.default_rng(4)
    for _ in range(100):
        frm = rng.uniform(-20, 20, 3); at = frm + rng.uniform(-10, 10, 3) + 0.5
        a = hip.camera_init(frm, at, (0, 1, 0), rng.uniform(20, 100), 1.5)
    frm, at = (50, 40, -120), (50, 40, 62)                     # the camera of main.cpp:56
    a = hip.camera_init(frm, at, (0, 1, 0), 95, 1020 / 630)
    b = orc.camera_init(frm, at, (0, 1, 0), 95, 1020 / 630)
    assert bytes(a) == bytes(b)


def test_bmp_header_and_payload(tmp_path):
    """Snapshot (bmpgenerator.cpp:3-45) layout with fixed-width fields."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (6, 8, 3), dtype=np.uint8)
    f = tmp_path / "t.bmp"
    hip.save_bmp(img, str(f))
    raw = f.read_bytes()
    assert len(raw) == 54 + 6 * 8 * 3 and raw[:2] == b"BM"
    assert int.from_bytes(raw[2:6], "little") == len(raw) and int.from_bytes(raw[10:14], "little") == 54
    assert int.from_bytes(raw[14:18], "little") == 40
    assert int.from_bytes(raw[18:22], "little") == 8 and int.from_bytes(raw[22:26], "little") == 6
    assert int.from_bytes(raw[26:28], "little") == 1 and int.from_bytes(raw[28:30], "little") == 24
    assert raw[54:] == img.tobytes()
    from PIL import Image
    im = np.asarray(Image.open(str(f)))                       # a standard reader agrees: bottom-up rows, BGR
    assert np.array_equal(im, img[::-1, :, ::-1])


@pytest.mark.parametrize("w", [7, 5, 2, 1])
def test_bmp_rows_are_padded_to_four_bytes(tmp_path, w):
    """Widths whose rows are not a multiple of four bytes: the file stays a valid 24-bit BMP (the
    reference's unpadded rows, bmpgenerator.cpp:24-41, would shear the image) and a standard
    reader returns the pixels."""
    from PIL import Image
    rng = np.random.default_rng(w)
    img = rng.integers(0, 256, (5, w, 3), dtype=np.uint8)
    f = tmp_path / "p.bmp"
    hip.save_bmp(img, str(f))
    raw = f.read_bytes()
    stride = (3 * w + 3) & ~3
    assert len(raw) == 54 + 5 * stride and int.from_bytes(raw[2:6], "little") == len(raw)
    assert int.from_bytes(raw[34:38], "little") == 5 * stride
    assert np.array_equal(np.asarray(Image.open(str(f))), img[::-1, :, ::-1])


def test_invalid_scenes_are_rejected_before_any_device_work():
    L = hip.lib()
    good, _ = scenes.three_sphere_scene()
    bad = D.SceneDesc(good.objects, good.materials, good.lights.copy(), [])
    bad.lights["type"][0] = 9
    with pytest.raises(hip.RtError, match="light type"):
        hip.HipScene(bad)
    bad = D.SceneDesc(good.objects, good.materials.copy(), good.lights, [])
    bad.materials["type"][2] = D.MAT_TRANSPARENT; bad.materials["nr"][2] = 0.0
    with pytest.raises(hip.RtError, match="nr"):
        hip.HipScene(bad)
    bad = D.SceneDesc(good.objects.copy(), good.materials, good.lights, [])
    bad.objects["material"][1] = 99
    with pytest.raises(hip.RtError, match="material"):
        hip.HipScene(bad)
    bad = D.SceneDesc(good.objects.copy(), good.materials, good.lights, [])
    bad.objects["type"][0] = D.OBJ_MESH; bad.objects["mesh"][0] = 3
    with pytest.raises(hip.RtError, match="mesh"):
        hip.HipScene(bad)
    bad = D.SceneDesc(good.objects, good.materials, good.lights, [], octree_max_depth=40)
    with pytest.raises(hip.RtError, match="octree_max_depth"):
        hip.HipScene(bad)
    tri = np.zeros((1, 9)); tri[0, 4] = np.nan
    bad = D.SceneDesc(scenes._stack([scenes.mesh_object(0, 0)], D.OBJECT_DT), good.materials, good.lights, [tri])
    with pytest.raises(hip.RtError, match="finite"):
        hip.HipScene(bad)
    assert L.rt_scene_create(None, None) != 0 and b"null" in L.rt_last_error()


def test_no_gpu_means_an_error_not_a_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    good, _ = scenes.three_sphere_scene()
    with pytest.raises(hip.RtError):
        hip.HipScene(good)


def test_product_does_not_touch_the_oracle():
    """Nothing under raytracer_amd/ or include/ may import, link or mention the oracle."""
    bad = []
    for base in ("raytracer_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".cpp", ".hip", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"\boracle\b|rt_oracle|\borc_", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
    out = subprocess.check_output(["ldd", hip.LIB_PATH]).decode()
    assert "oracle" not in out


def test_animated_camera_and_light_are_host_arithmetic_and_match_oracle():
    """ChangeFovCamera::timeChanged (camera.cpp:88-94) and MovableFaceLight (light.cpp:87-107)
    as host functions of the boundary: same bytes as the oracle's, closed form checked."""
    cam = dict(look_from=(1, 2, 9), look_at=(0, 1, 0), vup=(0, 1, 0), vfov=40.0, aspect=1.5)
    a0, b0 = hip.camera_init(**cam), orc.camera_init(**cam)
    assert bytes(a0) == bytes(b0) and a0.aperture == 0.0 and tuple(a0.look_at) == (0.0, 1.0, 0.0)
    t = 0.0
    for _ in range(140):                                   # the driver's clock, main.cpp:83
        a, b = hip.camera_change_fov(a0, t), orc.camera_change_fov(b0, t)
        assert bytes(a) == bytes(b)
        scale = np.sqrt(np.sin(t * 2 * 3.1415926) * 1.5 + 1.6) + 0.3
        assert a.width == a0.width * scale and a.height == a0.height * scale
        assert tuple(a.e) == tuple(a0.e) and tuple(a.u) == tuple(a0.u)
        centre = np.array(a.llc) + np.array(a.u) * a.width / 2 + np.array(a.v) * a.height / 2
        assert np.allclose(centre, a.look_at, atol=1e-12)
        t += 0.015
    l0 = scenes.face_light((-3, 6, 3), (0.4, -1, -0.3), (0, 0, 1), 2.5, 1.5, (.5, .5, .6))
    # the same along the driver's clock against what the reference's own classes gave
    with np.load(os.path.join(ROOT, "tests", "golden", "animation.npz")) as g:
        for t, gc, gp, gs in zip(g["times"], g["cams"], g["light_pos"], g["light_sample"]):
            c = hip.camera_change_fov(a0, float(t))
            assert np.array_equal(np.array(list(c.e) + list(c.llc) + list(c.u) + list(c.v) + [c.width, c.height]), gc)
            l = hip.light_move_face(l0, float(t))
            assert np.array_equal(l["pos"], gp)
            p, _, _ = orc.light_sample(orc.light_move_face(l0, float(t)), (0, 0, 0), 0.3, 0.8)
            assert np.array_equal(p, gs)
    for t in (0.0, 0.015, 0.25, 0.5, 1.37, 2.0):
        a, b = hip.light_move_face(l0, t), orc.light_move_face(l0, t)
        assert a.tobytes() == b.tobytes()
        r = np.sqrt(18.0)
        assert a["pos"][0] == np.sin(t * 2 * 3.1415926) * r and a["pos"][2] == np.cos(t * 2 * 3.1415926) * r
        assert a["pos"][1] == 6.0 and np.array_equal(a["normal"], l0["normal"]) and a["width"] == 2.5


def test_perlin_host_functions_match_oracle_and_golden():
    """rt_perlin_tables / rt_perlin_noise are host arithmetic: same tables as the oracle's replay
    of the generator, same noise as the values captured from the reference."""
    P1, G1 = hip.perlin_tables()
    P2, G2 = orc.perlin_tables()
    assert np.array_equal(P1, P2) and np.array_equal(G1, G2) and sorted(P1) == list(range(256))
    P3, G3 = hip.perlin_tables(seed=7, n=64)
    P4, G4 = orc.perlin_tables(seed=7, n=64)
    assert np.array_equal(P3, P4) and np.array_equal(G3, G4)
    with np.load(os.path.join(ROOT, "tests", "golden", "perlin.npz")) as g:
        for (x, y, z), sc, ref in zip(g["xyz"], g["scale"], g["noise"]):
            assert hip.perlin_noise(x, y, z, float(sc), g["P"], g["G"]) == ref
        # the vectors' drift over 20 frames of the driver's clock (algorithms.cpp:57-77)
        Ga, Gb = g["G"].copy(), g["G"].copy()
        ca, cb = hip.PerlinClockC(), hip.PerlinClockC()
        prev = t = 0.0
        for _ in range(20):
            hip.perlin_time_changed(ca, prev, t, Ga)
            orc.perlin_time_changed(cb, prev, t, Gb)
            prev, t = t, t + 0.015
        assert np.array_equal(Ga, g["G_after_20"]) and np.array_equal(Gb, g["G_after_20"])
        assert [ca.count, ca.stage, ca.stime] == list(g["clock_after_20"])


def test_avi_container_round_trip(tmp_path):
    """rt_avi_*: RIFF/AVI with uncompressed bottom-up 24-bit frames (the container AviSaver's
    OpenCV writer produces, AviSaver.cpp:57-64, minus the MJPG codec), parsed back chunk by chunk."""
    import struct
    rng = np.random.default_rng(3)
    W, H, N = 18, 7, 5                                     # 3*W = 54 is not a multiple of 4: rows are padded
    frames = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    f = str(tmp_path / "a.avi")
    with hip.AviWriter(f, W, H, 15.0) as avi:
        for fr in frames:
            avi.add(fr)
    raw = open(f, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"AVI " and struct.unpack_from("<I", raw, 4)[0] == len(raw) - 8
    pos, chunks = 12, {}
    while pos < len(raw):
        tag, size = raw[pos:pos + 4], struct.unpack_from("<I", raw, pos + 4)[0]
        chunks[(tag, raw[pos + 8:pos + 12] if tag == b"LIST" else b"")] = (pos + 8, size)
        pos += 8 + size + (size & 1)
    assert pos == len(raw)
    (hoff, hsize), (moff, msize), (ioff, isize) = chunks[(b"LIST", b"hdrl")], chunks[(b"LIST", b"movi")], chunks[(b"idx1", b"")]
    assert raw[hoff + 4:hoff + 8] == b"avih"
    us, _, _, flags, total, _, streams, _, w, h = struct.unpack_from("<10I", raw, hoff + 12)
    assert (us, total, streams, w, h) == (66667, N, 1, W, H) and flags & 0x10
    assert raw[hoff + 76:hoff + 84] == b"strlstrh" and raw[hoff + 88:hoff + 96] == b"vidsDIB "
    scale, rate = struct.unpack_from("<2I", raw, hoff + 88 + 20)
    assert rate / scale == 15.0
    stride = (3 * W + 3) & ~3
    pos = moff + 4
    for i in range(N):
        assert raw[pos:pos + 4] == b"00db" and struct.unpack_from("<I", raw, pos + 4)[0] == stride * H
        rows = np.frombuffer(raw, np.uint8, stride * H, pos + 8).reshape(H, stride)
        assert np.array_equal(rows[:, :3 * W].reshape(H, W, 3), frames[i])
        ck, fl, off, sz = struct.unpack_from("<4sIII", raw, ioff + 16 * i)
        assert ck == b"00db" and moff + off == pos and sz == stride * H
        pos += 8 + stride * H
    assert pos == moff + msize and isize == 16 * N
    with pytest.raises(hip.RtError):
        hip.AviWriter(str(tmp_path / "no_such_dir" / "b.avi"), W, H)


def test_mjpg_avi_frames_decode(tmp_path):
    """rt_avi_open_mjpg: every frame a baseline JPEG inside the AVI container; the frames are
    pulled out of the 'movi' list and decoded with an independent decoder (Pillow)."""
    import io
    import struct
    Image = pytest.importorskip("PIL.Image")
    W, H, N = 83, 61, 3                                    # not multiples of 8: edge blocks
    yy, xx = np.mgrid[0:H, 0:W]
    frames = []
    for k in range(N):
        img = np.zeros((H, W, 3), np.uint8)               # bottom-up B,G,R as resolve_bgr8 gives it
        img[..., 0] = (xx * 3 + 20 * k) % 256
        img[..., 1] = (yy * 4) % 256
        img[..., 2] = ((xx - 40) ** 2 + (yy - 30 - 5 * k) ** 2 < 300) * 220
        frames.append(img)
    f = str(tmp_path / "m.avi")
    with hip.AviWriter(f, W, H, 15.0, mjpg_quality=92) as avi:
        for fr in frames:
            avi.add(fr)
    raw = open(f, "rb").read()
    assert raw[:4] == b"RIFF" and struct.unpack_from("<I", raw, 4)[0] == len(raw) - 8
    assert raw.count(b"vidsMJPG") == 1 and b"00dc" in raw
    movi = raw.index(b"movi")
    idx = raw.rindex(b"idx1")
    assert struct.unpack_from("<I", raw, idx + 4)[0] == 16 * N
    for k in range(N):
        ck, fl, off, sz = struct.unpack_from("<4sIII", raw, idx + 8 + 16 * k)
        assert ck == b"00dc" and raw[movi + off:movi + off + 4] == b"00dc"
        jpg = raw[movi + off + 8:movi + off + 8 + sz]
        assert jpg[:2] == b"\xff\xd8" and jpg[-2:] == b"\xff\xd9"
        dec = np.asarray(Image.open(io.BytesIO(jpg)).convert("RGB")).astype(float)       # top-down R,G,B
        src = frames[k][::-1, :, ::-1].astype(float)
        assert dec.shape == src.shape
        psnr = 10 * np.log10(255.0 ** 2 / ((dec - src) ** 2).mean())
        assert psnr > 30, psnr
    assert len(raw) < N * W * H * 3                                                      # it does compress
