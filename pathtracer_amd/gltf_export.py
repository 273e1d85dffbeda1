"""Writes a `Scene` as a binary glTF (.glb) the reference's loader conventions can read back.

The reference ships no assets, so BASELINE.json's glTF configurations
(cube.glb, the Cornell box, the textured scene) are generated: scenes.py builds the
geometry, this module serialises it, and the C++ front end (csrc/host/gltf_loader.cpp, the
restatement of src/gltf_loader/gltf_loader.cc) loads it.  The export inverts the loader's
index conventions (Q17): material / image / sampler 0 are the loader's built-ins and are
not written, gltf material i is scene material i + 1, texture indices are not shifted,
the trailing dummy light of each array is dropped, light intensities are divided by the
4*pi/683 factor (Q9).  Vertices are written under an identity node so positions, uvs and
material indices come back bit for bit; normals are re-normalised by the loader.
"""
from __future__ import annotations

import json
import math
import struct
import zlib

import numpy as np

from . import abi
from .scene import Scene

FILTER_GL = {abi.FILTER_NEAREST: 9728, abi.FILTER_LINEAR: 9729}
WRAP_GL = {abi.WRAP_CLAMP_TO_EDGE: 33071, abi.WRAP_MIRRORED_REPEAT: 33648, abi.WRAP_REPEAT: 10497}


def encode_png(rgba: np.ndarray, filter_type: int = 0) -> bytes:
    """RGBA8 PNG (colour type 6, depth 8), one filter type for every row."""
    img = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w, c = img.shape
    assert c == 4
    rows = img.reshape(h, w * 4).astype(np.int32)
    if filter_type == 0:
        filt = rows
    elif filter_type == 1:  # Sub
        left = np.concatenate([np.zeros((h, 4), np.int32), rows[:, :-4]], axis=1)
        filt = rows - left
    elif filter_type == 2:  # Up
        up = np.concatenate([np.zeros((1, w * 4), np.int32), rows[:-1]], axis=0)
        filt = rows - up
    else:
        raise ValueError("filter type 0..2 only")
    raw = np.concatenate([np.full((h, 1), filter_type, np.uint8), (filt & 255).astype(np.uint8)], axis=1).tobytes()

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def _quat_from_minus_z(d) -> list[float]:
    """Unit quaternion (x, y, z, w) rotating (0, 0, -1) onto direction d."""
    d = np.asarray(d, dtype=np.float64)
    d = d / np.linalg.norm(d)
    f = np.array([0.0, 0.0, -1.0])
    c = float(np.dot(f, d))
    if c < -1.0 + 1e-12:  # opposite: half turn about y
        return [0.0, 1.0, 0.0, 0.0]
    axis = np.cross(f, d)
    q = np.array([axis[0], axis[1], axis[2], 1.0 + c])
    q /= np.linalg.norm(q)
    return [float(x) for x in q]


class _Bin:
    def __init__(self):
        self.data = bytearray()
        self.views = []
        self.accessors = []

    def view(self, payload: bytes, target: int | None = None) -> int:
        while len(self.data) % 4:
            self.data.append(0)
        v = {"buffer": 0, "byteOffset": len(self.data), "byteLength": len(payload)}
        if target:
            v["target"] = target
        self.data += payload
        self.views.append(v)
        return len(self.views) - 1

    def accessor(self, arr: np.ndarray, ctype: int, gtype: str, target: int, minmax: bool = False) -> int:
        a = np.ascontiguousarray(arr)
        acc = {"bufferView": self.view(a.tobytes(), target), "componentType": ctype, "count": int(a.shape[0]), "type": gtype}
        if gtype == "SCALAR":
            acc["count"] = int(a.size)
        if minmax:
            acc["min"] = [float(x) for x in a.min(axis=0)]
            acc["max"] = [float(x) for x in a.max(axis=0)]
        self.accessors.append(acc)
        return len(self.accessors) - 1


def scene_to_glb(scene: Scene, with_camera: bool = True, index_type: str = "auto", png_filter: int = 1) -> bytes:
    """Serialise `scene`: one mesh primitive per material (first-use order), identity node."""
    b = _Bin()
    verts = scene.vertices
    tris = np.asarray(scene.triangles, dtype=np.uint32)
    tri_mat = verts["material_index"][tris[:, 0]]
    primitives = []
    # primitives in order of first use, so the loader's concatenation keeps a stable order
    order = []
    for m in tri_mat:
        if int(m) not in order:
            order.append(int(m))
    for m in order:
        sel = tris[tri_mat == m]
        used, inverse = np.unique(sel.reshape(-1), return_inverse=True)
        if not (verts["material_index"][used] == m).all():
            raise ValueError("a triangle mixes vertices of different materials; cannot export as primitives")
        local = inverse.reshape(-1).astype(np.uint32)
        if index_type == "u16" or (index_type == "auto" and len(used) < 65536):
            idx = b.accessor(local.astype(np.uint16), 5123, "SCALAR", 34963)
        else:
            idx = b.accessor(local, 5125, "SCALAR", 34963)
        prim = {"mode": 4, "indices": idx, "attributes": {
            "POSITION": b.accessor(verts["position"][used].astype(np.float32), 5126, "VEC3", 34962, minmax=True),
            "NORMAL": b.accessor(verts["normal"][used].astype(np.float32), 5126, "VEC3", 34962),
            "TEXCOORD_0": b.accessor(verts["texcoord"][used].astype(np.float32), 5126, "VEC2", 34962),
        }}
        if m > 0:
            prim["material"] = m - 1
        primitives.append(prim)

    doc = {"asset": {"version": "2.0", "generator": "pathtracer_amd.gltf_export"}, "scene": 0,
           "meshes": [{"primitives": primitives}]}
    nodes = [{"mesh": 0, "name": "geometry"}]

    def tex_ref(index: int) -> dict | None:
        return None if index < 0 else {"index": int(index)}

    mats = []
    for m in scene.materials[1:]:
        pbr = {"baseColorFactor": [float(x) for x in m["base_color_factor"]],
               "metallicFactor": float(m["metallic_factor"]), "roughnessFactor": float(m["roughness_factor"])}
        if (t := tex_ref(int(m["base_color_texture_index"]))):
            pbr["baseColorTexture"] = t
        if (t := tex_ref(int(m["metallic_roughness_texture_index"]))):
            pbr["metallicRoughnessTexture"] = t
        gm = {"pbrMetallicRoughness": pbr}
        ef = np.asarray(m["emissive_factor"], dtype=np.float64)
        peak = float(ef.max())
        if peak > 1.0:  # glTF caps emissiveFactor at 1: carry the scale in KHR_materials_emissive_strength
            # strength is a power of two so that float(factor) * float(strength) restores the bits
            strength = float(2.0 ** math.ceil(math.log2(peak)))
            gm["emissiveFactor"] = [float(np.float32(x / strength)) for x in ef]
            gm["extensions"] = {"KHR_materials_emissive_strength": {"emissiveStrength": strength}}
        elif peak > 0.0:
            gm["emissiveFactor"] = [float(x) for x in ef]
        if (t := tex_ref(int(m["emissive_texture_index"]))):
            gm["emissiveTexture"] = t
        if (t := tex_ref(int(m["occlusion_texture_index"]))):
            t["strength"] = float(m["occlusion_strength"])
            gm["occlusionTexture"] = t
        if (t := tex_ref(int(m["normal_texture_index"]))):
            t["scale"] = float(m["normal_scale"])
            gm["normalTexture"] = t
        mats.append(gm)
    if mats:
        doc["materials"] = mats

    images = []
    for img in scene.images[1:]:
        images.append({"bufferView": b.view(encode_png(img, png_filter)), "mimeType": "image/png"})
    if images:
        doc["images"] = images
    if len(scene.samplers) > 1:
        doc["samplers"] = [{"magFilter": FILTER_GL[int(s["mag_filter"])], "minFilter": FILTER_GL[int(s["min_filter"])],
                            "wrapS": WRAP_GL[int(s["wrap_s"])], "wrapT": WRAP_GL[int(s["wrap_t"])]} for s in scene.samplers[1:]]
    uses_textures = any(int(m[k]) >= 0 for m in scene.materials for k in (
        "base_color_texture_index", "metallic_roughness_texture_index", "emissive_texture_index", "occlusion_texture_index"))
    if uses_textures:
        texs = []
        for t in scene.textures:
            g = {}
            if int(t["sampler_index"]) > 0:
                g["sampler"] = int(t["sampler_index"]) - 1
            if int(t["image_index"]) > 0:
                g["source"] = int(t["image_index"]) - 1
            texs.append(g)
        doc["textures"] = texs

    lights = []
    scale = 4.0 * float(np.float32(math.pi)) / 683.0
    for l in scene.directional_lights[:-1]:
        lights.append({"type": "directional", "color": [float(x) for x in l["color"]], "intensity": float(l["intensity"]) / scale})
        nodes.append({"name": f"dir{len(lights)}", "rotation": _quat_from_minus_z(l["direction"]),
                      "extensions": {"KHR_lights_punctual": {"light": len(lights) - 1}}})
    for l in scene.point_lights[:-1]:
        lights.append({"type": "point", "color": [float(x) for x in l["color"]], "intensity": float(l["intensity"]) / scale})
        nodes.append({"name": f"point{len(lights)}", "translation": [float(x) for x in l["position"]],
                      "extensions": {"KHR_lights_punctual": {"light": len(lights) - 1}}})
    used_ext = set()
    if lights:
        doc["extensions"] = {"KHR_lights_punctual": {"lights": lights}}
        used_ext.add("KHR_lights_punctual")
    if any("extensions" in m for m in mats):
        used_ext.add("KHR_materials_emissive_strength")
    if used_ext:
        doc["extensionsUsed"] = sorted(used_ext)

    if with_camera:
        doc["cameras"] = [{"type": "perspective", "perspective": {"yfov": float(scene.camera_vfov), "znear": 0.01}}]
        nodes.append({"name": "camera", "camera": 0, "translation": [float(x) for x in scene.camera_position],
                      "rotation": _quat_from_minus_z(scene.camera_lookat)})
    doc["nodes"] = nodes
    doc["scenes"] = [{"nodes": list(range(len(nodes)))}]
    doc["bufferViews"] = b.views
    doc["accessors"] = b.accessors
    while len(b.data) % 4:
        b.data.append(0)
    doc["buffers"] = [{"byteLength": len(b.data)}]

    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * (-len(js) % 4)
    total = 12 + 8 + len(js) + 8 + len(b.data)
    return (struct.pack("<4sII", b"glTF", 2, total) + struct.pack("<I4s", len(js), b"JSON") + js +
            struct.pack("<I4s", len(b.data), b"BIN\0") + bytes(b.data))


def write_glb(scene: Scene, path: str, **kw) -> None:
    with open(path, "wb") as f:
        f.write(scene_to_glb(scene, **kw))
