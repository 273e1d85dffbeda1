"""The C++ front end (glTF/GLB loader, PNG codec, CLI) against generated assets.

The reference ships no assets and no loader tests; tinygltf / glm / stb are
absent, so the loader's last-bit agreement with glm's matrix helpers is "parity unpinned".
What is pinned here is the BEHAVIOUR SURVEY.md Q9/Q17/Q18 records from gltf_loader.cc:
index conventions, defaults, light scaling, auto-camera, u16/u32 indices, node transforms.
"""
import base64
import ctypes as C
import json
import math
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from pathtracer_amd import abi, frontend, gltf_export, scenes
from pathtracer_amd.scene import default_image, default_material

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tri_key(scene):
    """Order-independent fingerprint of the triangle soup: per-triangle vertex records."""
    v = scene.vertices[np.asarray(scene.triangles).reshape(-1)]
    rec = np.concatenate([v["position"], v["texcoord"], v["material_index"][:, None].astype(np.float32)], axis=1)
    return rec.reshape(len(scene.triangles), -1)


@pytest.mark.parametrize("name", ["cube", "cornell", "textured"])
def test_glb_round_trip(tmp_path, ptlib, name):
    src = {"cube": scenes.cube_scene, "cornell": lambda: scenes.cornell_scene(True, True, True, True),
           "textured": scenes.textured_scene}[name]()
    path = tmp_path / f"{name}.glb"
    gltf_export.write_glb(src, str(path), with_camera=(name != "cube"))
    got = frontend.load_gltf(str(path))

    assert len(got.triangles) == len(src.triangles)
    a, b = tri_key(src), tri_key(got)
    # primitives are grouped by material on export: compare as sets of triangles, bit for bit
    assert sorted(map(bytes, a)) == sorted(map(bytes, b))
    # normals: the loader re-normalises (gltf_loader.cc:397)
    n = src.vertices["normal"].astype(np.float32)
    inv = np.float32(1.0) / np.sqrt((n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1]) + n[:, 2] * n[:, 2])
    want = np.sort((n * inv[:, None]).view(np.uint32).reshape(-1))
    assert np.array_equal(want, np.sort(got.vertices["normal"].view(np.uint32).reshape(-1)))

    assert got.materials.tobytes() == src.materials.tobytes()
    assert got.textures.tobytes() == src.textures.tobytes()
    assert got.samplers.tobytes() == src.samplers.tobytes()
    assert len(got.images) == len(src.images) and all(np.array_equal(x, y) for x, y in zip(src.images, got.images))
    # Q17: trailing dummy light kept at the end of each array
    assert len(got.directional_lights) == len(src.directional_lights) and len(got.point_lights) == len(src.point_lights)
    assert not got.directional_lights[-1].tobytes().strip(b"\0") and not got.point_lights[-1].tobytes().strip(b"\0")
    for k in ("color", "intensity", "direction"):
        assert np.allclose(got.directional_lights[k], src.directional_lights[k], rtol=2e-6, atol=1e-7)
    for k in ("color", "intensity", "position"):
        assert np.allclose(got.point_lights[k], src.point_lights[k], rtol=2e-6, atol=1e-7)
    assert np.allclose(got.camera_position, src.camera_position, rtol=1e-6)
    assert np.allclose(got.camera_lookat, src.camera_lookat, atol=2e-6)
    assert got.camera_vfov == pytest.approx(src.camera_vfov, rel=1e-7)


def test_auto_camera_and_defaults_q17_q18(tmp_path, ptlib):
    src = scenes.cube_scene()
    path = tmp_path / "cube.glb"
    gltf_export.write_glb(src, str(path), with_camera=False)
    got = frontend.load_gltf(str(path))
    # no cameras -> 1.75 x max corner, looking at the origin, 45 degrees (gltf_loader.cc:457-471)
    assert got.camera_position == (1.75, 1.75, 1.75)
    assert np.array(got.camera_lookat, np.float32).tobytes() == np.array(src.camera_lookat, np.float32).tobytes()
    assert np.float32(got.camera_vfov) == np.float32(math.radians(45.0))
    # material 0 / image 0 / sampler 0 are the built-ins; one dummy texture {0, 0}
    assert got.materials.tobytes() == default_material().tobytes()
    assert len(got.images) == 1 and np.array_equal(got.images[0], default_image())
    assert got.samplers.tolist() == [(abi.FILTER_LINEAR, abi.FILTER_LINEAR, abi.WRAP_REPEAT, abi.WRAP_REPEAT)]
    assert got.textures.tolist() == [(0, 0)]
    assert (got.vertices["material_index"] == 0).all()


def _glb(doc: dict, bin_data: bytes) -> bytes:
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    bin_data += b"\0" * (-len(bin_data) % 4)
    return (struct.pack("<4sII", b"glTF", 2, 28 + len(js) + len(bin_data)) + struct.pack("<I4s", len(js), b"JSON") + js +
            struct.pack("<I4s", len(bin_data), b"BIN\0") + bin_data)


def _one_triangle_doc(index_ctype=5123, node=None, extra=None):
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    nrm = np.array([[0, 0, 1]] * 3, np.float32)
    uv = np.array([[0, 0], [1, 0], [0, 1]], np.float32)
    idx = np.array([0, 1, 2], np.uint16 if index_ctype == 5123 else (np.uint32 if index_ctype == 5125 else np.uint8))
    blobs = [pos.tobytes(), nrm.tobytes(), uv.tobytes(), idx.tobytes() + b"\0" * (-idx.nbytes % 4)]
    views, off = [], 0
    for b in blobs:
        views.append({"buffer": 0, "byteOffset": off, "byteLength": len(b)})
        off += len(b)
    doc = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
        "nodes": [dict({"mesh": 0}, **(node or {}))],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3}]}],
        "accessors": [
            {"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
            {"bufferView": 1, "componentType": 5126, "count": 3, "type": "VEC3"},
            {"bufferView": 2, "componentType": 5126, "count": 3, "type": "VEC2"},
            {"bufferView": 3, "componentType": index_ctype, "count": 3, "type": "SCALAR"},
        ],
        "bufferViews": views, "buffers": [{"byteLength": off}],
    }
    if extra:
        doc.update(extra)
    return doc, b"".join(blobs)


def test_node_trs_transform(tmp_path, ptlib):
    # T(1,2,3) * R(90 deg about +y) * S(2): (x,y,z) -> (2z + 1, 2y + 2, -2x + 3)
    h = math.sqrt(0.5)
    doc, blob = _one_triangle_doc(node={"translation": [1, 2, 3], "rotation": [0, h, 0, h], "scale": [2, 2, 2]})
    p = tmp_path / "t.glb"
    p.write_bytes(_glb(doc, blob))
    s = frontend.load_gltf(str(p))
    want = np.array([[1, 2, 3], [1, 2, 1], [1, 4, 3]], np.float32)
    assert np.allclose(s.vertices["position"], want, atol=1e-6)
    assert np.allclose(s.vertices["normal"], [[1, 0, 0]] * 3, atol=1e-6)  # (0,0,1) rotated onto +x, unit length
    assert s.triangles.tolist() == [[0, 1, 2]]


def test_node_matrix_and_children_accumulate(tmp_path, ptlib):
    doc, blob = _one_triangle_doc()
    m = np.eye(4)
    m[:3, 3] = (5, 0, 0)
    doc["nodes"] = [{"matrix": m.T.reshape(-1).tolist(), "children": [1]}, {"mesh": 0, "translation": [0, 1, 0]}]
    p = tmp_path / "m.glb"
    p.write_bytes(_glb(doc, blob))
    s = frontend.load_gltf(str(p))
    assert np.array_equal(s.vertices["position"], np.array([[5, 1, 0], [6, 1, 0], [5, 2, 0]], np.float32))


def test_u32_indices_and_two_primitives_offset(tmp_path, ptlib):
    doc, blob = _one_triangle_doc(index_ctype=5125)
    doc["meshes"][0]["primitives"] *= 2
    p = tmp_path / "u32.glb"
    p.write_bytes(_glb(doc, blob))
    s = frontend.load_gltf(str(p))
    assert s.triangles.tolist() == [[0, 1, 2], [3, 4, 5]] and len(s.vertices) == 6  # index_offset, gltf_loader.cc:425


def test_u8_indices_rejected_like_reference(tmp_path, ptlib):
    doc, blob = _one_triangle_doc(index_ctype=5121)
    p = tmp_path / "u8.glb"
    p.write_bytes(_glb(doc, blob))
    with pytest.raises(RuntimeError, match="Unknown index type"):
        frontend.load_gltf(str(p))


def test_lights_intensity_scale_q9_and_json_gltf_with_data_uri(tmp_path, ptlib):
    ext = {"extensions": {"KHR_lights_punctual": {"lights": [
        {"type": "point", "color": [1, 0.5, 0.25], "intensity": 683.0},
        {"type": "directional", "intensity": 100.0},
        {"type": "spot", "intensity": 5.0}]}}}
    doc, blob = _one_triangle_doc(extra=ext)
    doc["nodes"] += [
        {"translation": [1, 2, 3], "extensions": {"KHR_lights_punctual": {"light": 0}}},
        {"extensions": {"KHR_lights_punctual": {"light": 1}}},
        {"extensions": {"KHR_lights_punctual": {"light": 2}}},
    ]
    doc["scenes"][0]["nodes"] = [0, 1, 2, 3]
    doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(blob).decode()
    p = tmp_path / "lights.gltf"
    p.write_text(json.dumps(doc))
    s = frontend.load_gltf(str(p))
    scale = 4.0 * float(np.float32(math.pi)) / 683.0  # double arithmetic on pi<float>() (gltf_loader.cc:431)
    assert len(s.point_lights) == 2 and len(s.directional_lights) == 2  # spot ignored; one dummy each
    assert s.point_lights[0]["intensity"] == np.float32(683.0 * scale)
    assert s.point_lights[0]["position"].tolist() == [1, 2, 3] and s.point_lights[0]["color"].tolist() == [1, 0.5, 0.25]
    assert s.directional_lights[0]["intensity"] == np.float32(100.0 * scale)
    assert s.directional_lights[0]["direction"].tolist() == [0, 0, -1] and s.directional_lights[0]["color"].tolist() == [1, 1, 1]


def test_external_files_with_percent_encoded_names(tmp_path, ptlib):
    """.gltf with its buffer and an image in files of their own, named with spaces and written percent-encoded in the JSON
    (what exporters do; tinygltf decodes the URI before it opens the file)."""
    from pathtracer_amd.gltf_export import encode_png
    doc, blob = _one_triangle_doc(extra={
        "images": [{"uri": "base%20colour%2B1.png"}], "samplers": [{}], "textures": [{"source": 0, "sampler": 0}],
        "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}]})
    doc["meshes"][0]["primitives"][0]["material"] = 0
    doc["buffers"][0]["uri"] = "tri%20angle.bin"
    (tmp_path / "tri angle.bin").write_bytes(blob)
    tex = np.random.RandomState(2).randint(0, 256, size=(3, 5, 4)).astype(np.uint8)
    (tmp_path / "base colour+1.png").write_bytes(encode_png(tex))
    p = tmp_path / "ext.gltf"
    p.write_text(json.dumps(doc))
    s = frontend.load_gltf(str(p))
    assert len(s.triangles) == 1 and len(s.images) == 2 and np.array_equal(s.images[1], tex)
    doc["buffers"][0]["uri"] = "missing%20file.bin"
    p.write_text(json.dumps(doc))
    with pytest.raises(RuntimeError):
        frontend.load_gltf(str(p))


def test_material_defaults_and_emissive_strength(tmp_path, ptlib):
    doc, blob = _one_triangle_doc(extra={"materials": [
        {},
        {"pbrMetallicRoughness": {"baseColorFactor": [0.1, 0.2, 0.3, 0.5], "metallicFactor": 0.25, "roughnessFactor": 0.75},
         "emissiveFactor": [0.5, 0.25, 1.0], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 8.0}}}]})
    doc["meshes"][0]["primitives"][0]["material"] = 1
    p = tmp_path / "mat.glb"
    p.write_bytes(_glb(doc, blob))
    s = frontend.load_gltf(str(p))
    assert len(s.materials) == 3 and (s.vertices["material_index"] == 2).all()  # gltf material i -> i + 1
    d = s.materials[1]  # tinygltf defaults: white, metal 1, rough 1, no textures, no emission
    assert d["base_color_factor"].tolist() == [1, 1, 1, 1] and d["metallic_factor"] == 1 and d["roughness_factor"] == 1
    assert d["base_color_texture_index"] == -1 and d["emissive_texture_index"] == -1 and d["emissive_factor"].tolist() == [0, 0, 0]
    m = s.materials[2]
    assert np.allclose(m["base_color_factor"], [0.1, 0.2, 0.3, 0.5]) and m["metallic_factor"] == 0.25
    assert m["emissive_factor"].tolist() == [4.0, 2.0, 8.0]


def test_loader_errors(tmp_path, ptlib):
    with pytest.raises(RuntimeError, match="Failed to open"):
        frontend.load_gltf(str(tmp_path / "missing.glb"))
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"not a glb at all, sorry")
    with pytest.raises(RuntimeError, match="Failed to parse glTF"):
        frontend.load_gltf(str(bad))
    doc, blob = _one_triangle_doc()
    doc["accessors"][0]["count"] = 1000  # reads past the buffer
    p = tmp_path / "oob.glb"
    p.write_bytes(_glb(doc, blob))
    with pytest.raises(RuntimeError, match="past the end"):
        frontend.load_gltf(str(p))
    doc, blob = _one_triangle_doc()
    doc["accessors"][3]["count"] = 3
    blob = blob[:-8] + np.array([0, 1, 7], np.uint16).tobytes() + b"\0\0"
    p = tmp_path / "idx.glb"
    p.write_bytes(_glb(doc, blob))
    with pytest.raises(RuntimeError, match="vertex index out of range"):
        frontend.load_gltf(str(p))


# ------------------------------------------------------------------------------ PNG codec
def _png(width, height, depth, color, rows: bytes, extra_chunks=b"", interlace=0):
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, color, 0, 0, interlace)) +
            extra_chunks + chunk(b"IDAT", zlib.compress(rows)) + chunk(b"IEND", b""))


def test_png_decode_rgba_filters_and_write_round_trip(tmp_path, ptlib):
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, size=(13, 17, 4)).astype(np.uint8)
    for f in (0, 1, 2):
        assert np.array_equal(frontend.decode_image(gltf_export.encode_png(img, f)), img)
    path = tmp_path / "w.png"
    frontend.write_png(str(path), img)  # stbi_write_png stand-in (main.cc:78-83)
    assert np.array_equal(frontend.decode_image(path.read_bytes()), img)


def test_png_decode_average_and_paeth_filters(ptlib):
    rng = np.random.RandomState(2)
    img = rng.randint(0, 256, size=(6, 5, 4)).astype(np.int32)
    h, w = img.shape[:2]
    flat = img.reshape(h, w * 4)
    for ftype in (3, 4):
        rows = bytearray()
        for y in range(h):
            rows.append(ftype)
            for i in range(w * 4):
                a = flat[y, i - 4] if i >= 4 else 0
                b = flat[y - 1, i] if y else 0
                c = flat[y - 1, i - 4] if (y and i >= 4) else 0
                if ftype == 3:
                    pred = (a + b) // 2
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                rows.append((flat[y, i] - pred) & 255)
        got = frontend.decode_image(_png(w, h, 8, 6, bytes(rows)))
        assert np.array_equal(got, img.astype(np.uint8))


def test_png_decode_grey_rgb_palette_and_16bit(ptlib):
    # grey 8-bit -> replicated, alpha 255
    g = np.arange(12, dtype=np.uint8).reshape(3, 4) * 20
    rows = b"".join(b"\0" + g[y].tobytes() for y in range(3))
    out = frontend.decode_image(_png(4, 3, 8, 0, rows))
    assert np.array_equal(out[..., 0], g) and np.array_equal(out[..., 2], g) and (out[..., 3] == 255).all()
    # RGB 8-bit
    rgb = np.random.RandomState(3).randint(0, 256, size=(2, 3, 3)).astype(np.uint8)
    rows = b"".join(b"\0" + rgb[y].tobytes() for y in range(2))
    out = frontend.decode_image(_png(3, 2, 8, 2, rows))
    assert np.array_equal(out[..., :3], rgb) and (out[..., 3] == 255).all()
    # palette, 2 bits per pixel, with tRNS
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    pal = bytes([255, 0, 0, 0, 255, 0, 0, 0, 255, 9, 8, 7])
    extra = chunk(b"PLTE", pal) + chunk(b"tRNS", bytes([255, 128]))
    out = frontend.decode_image(_png(4, 1, 2, 3, b"\0" + bytes([0b00011011]), extra))
    assert out[0].tolist() == [[255, 0, 0, 255], [0, 255, 0, 128], [0, 0, 255, 255], [9, 8, 7, 255]]
    # RGBA 16-bit keeps the high byte
    px = struct.pack(">HHHH", 0x1234, 0xABCD, 0x00FF, 0xFFFF)
    out = frontend.decode_image(_png(1, 1, 16, 6, b"\0" + px))
    assert out[0, 0].tolist() == [0x12, 0xAB, 0x00, 0xFF]


def test_baseline_jpeg_decode_against_libjpeg(ptlib):
    """Baseline JPEG textures (what the reference reads through tinygltf / stb_image). A lossy codec's pixels are
    decoder-defined (IDCT rounding, chroma upsampling): this decoder stays within 3 code values of libjpeg (PIL) on
    4:4:4, 4:2:2, 4:2:0, greyscale and restart-interval streams; parity with stb_image's exact bits is unpinned."""
    import io
    from PIL import Image
    y, x = np.mgrid[0:67, 0:93]
    img = np.stack([128 + 100 * np.sin(x / 9.0) * np.cos(y / 7.0), 128 + 90 * np.cos(x / 5.0 + y / 11.0), (x * 2 + y) % 256], -1)
    img = img.clip(0, 255).astype(np.uint8)
    cases = [dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=2, restart_marker_blocks=3),
             dict(subsampling=0, quality=35), None]
    for kw in cases:
        b = io.BytesIO()
        if kw is None:
            Image.fromarray(img[..., 0]).save(b, "JPEG", quality=90)
        else:
            Image.fromarray(img).save(b, "JPEG", **{"quality": 90, **kw})
        data = b.getvalue()
        ref = np.array(Image.open(io.BytesIO(data)).convert("RGB")).astype(int)
        got = frontend.decode_image(data)
        assert got.shape == (67, 93, 4) and (got[..., 3] == 255).all()
        d = np.abs(got[..., :3].astype(int) - ref)
        assert d.max() <= 3 and d.mean() < 0.5, (kw, d.max(), d.mean())

def test_progressive_jpeg_decode_against_libjpeg(ptlib):
    """Progressive JPEG (SOF2: spectral selection + successive approximation, what stb_image also reads). libjpeg's
    default progression has DC first / refinement scans and AC first / refinement scans with end-of-band runs; smooth and
    noisy content, three chroma layouts, greyscale, restart intervals, low quality (long zero runs and EOB runs). Same bar
    as the baseline test (decoder-defined rounding), and the coefficients of a progressive stream are those of the baseline
    stream libjpeg writes for the same pixels, so both must decode to the same bytes here."""
    import io
    from PIL import Image
    y, x = np.mgrid[0:131, 0:150]
    smooth = np.stack([128 + 100 * np.sin(x / 9.0) * np.cos(y / 7.0), 128 + 90 * np.cos(x / 5.0 + y / 11.0), (x * 2 + y) % 256], -1)
    noisy = smooth + np.random.RandomState(5).normal(0, 25, smooth.shape)
    for name, src in (("smooth", smooth), ("noisy", noisy)):
        img = src.clip(0, 255).astype(np.uint8)
        for kw in (dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=2, restart_marker_blocks=5),
                   dict(subsampling=0, quality=20), dict(subsampling=2, quality=98), None):
            prog, base = io.BytesIO(), io.BytesIO()
            pil = Image.fromarray(img[..., 0]) if kw is None else Image.fromarray(img)
            args = {"quality": 90, **(kw or {})}
            pil.save(prog, "JPEG", progressive=True, **args)
            pil.save(base, "JPEG", **args)
            data = prog.getvalue()
            assert b"\xff\xc2" in data and b"\xff\xc2" not in base.getvalue()
            ref = np.array(Image.open(io.BytesIO(data)).convert("RGB")).astype(int)
            got = frontend.decode_image(data)
            assert got.shape == (131, 150, 4) and (got[..., 3] == 255).all()
            d = np.abs(got[..., :3].astype(int) - ref)
            assert d.max() <= 3 and d.mean() < 0.5, (name, kw, d.max(), d.mean())
            assert np.array_equal(got, frontend.decode_image(base.getvalue())), (name, kw)
    arith = bytearray(data)
    arith[data.index(b"\xff\xc2") + 1] = 0xCA  # SOF10, progressive arithmetic-coded: refused by name, not mis-decoded
    with pytest.raises(RuntimeError):
        frontend.decode_image(bytes(arith))


def test_truncated_jpeg_and_garbage_rejected(ptlib):
    with pytest.raises(RuntimeError):
        frontend.decode_image(b"\xff\xd8\xff\xe0" + b"\0" * 32)
    with pytest.raises(RuntimeError):
        frontend.decode_image(b"hello world, not an image")


# ------------------------------------------------------------------------------ CLI
CLI = os.path.join(ROOT, "pathtracer_amd", "pathtracer")


def test_cli_reports_errors_like_the_reference(tmp_path):
    # main.cc:86-90: exceptions are printed and the exit code stays 0
    if not os.path.exists(CLI):
        pytest.skip("CLI not built")
    out = subprocess.run([CLI, str(tmp_path / "nope.json")], capture_output=True, text=True, cwd=tmp_path, timeout=60)
    assert out.returncode == 0 and "Exception Occured" in out.stdout


@pytest.mark.gpu
def test_cli_renders_the_readme_config(tmp_path, oracle):
    """pathtracer config.json end to end (main.cc:12-91): glb -> BVH -> upload -> render -> PNG,
    compared with the CPU oracle fed by the same loader output."""
    import oracle_lib
    from pathtracer_amd.scene import PackedScene
    glb = tmp_path / "cornell.glb"
    gltf_export.write_glb(scenes.cornell_scene(with_point_light=True), str(glb))
    cfg = {"resolution_x": 96, "resolution_y": 54, "sample_count": 16, "bg_color": "#3D3D3D",
           "gltf_path": str(glb), "seed": 77, "stats": True}
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    out = subprocess.run([CLI], capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert out.returncode == 0 and "Exception" not in out.stdout, out.stdout + out.stderr
    for label in ("GLTF Parsing", "BVH Construction", "Rendering", "Saving PNG"):
        assert label in out.stdout
    img = frontend.decode_image((tmp_path / "cornell.png").read_bytes())
    assert img.shape == (54, 96, 4)
    scene = frontend.load_gltf(str(glb))
    packed = PackedScene(scene, oracle, prefix="orc_")
    bg = tuple(np.float32(0x3D) / np.float32(255.0) for _ in range(3))
    rp = packed.render_params(oracle, 96, 54, 16, bg, 77, prefix="orc_")
    o_rgba = oracle_lib.render(oracle, packed, rp, threads=4)[0]
    assert np.array_equal(img, o_rgba)


@pytest.mark.gpu
def test_atrium_through_the_glb_front_end(tmp_path, oracle):
    """cfg3's generator at low tessellation, written as .glb (PNG textures embedded), read back by the C++ loader
    and rendered by the HIP path: equal to the oracle fed by the same loaded scene."""
    import oracle_lib
    from pathtracer_amd.renderer import Renderer
    glb = tmp_path / "atrium.glb"
    gltf_export.write_glb(scenes.atrium_scene(detail=2, tex_size=32), str(glb))
    scene = frontend.load_gltf(str(glb))
    assert len(scene.images) == 1 + 2 * 22 + 1 and len(scene.directional_lights) == 2
    r = Renderer(0)
    try:
        packed = r.pack(scene)
        r.upload(packed)
        rp = packed.render_params(r.lib, 96, 54, 9, (0.5, 0.7, 1.0), 41)
        rgba, accum, ids, st = r.render_params(rp)
    finally:
        r.close()
    o_rgba, o_accum, o_ids, _ = oracle_lib.render(oracle, packed, rp, threads=8)
    assert np.array_equal(ids, o_ids) and np.array_equal(accum.view(np.uint32), o_accum.view(np.uint32)) and np.array_equal(rgba, o_rgba)
    assert st["texture_taps"] > 0
