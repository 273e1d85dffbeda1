#!/bin/bash
# Damaged-file fuzz of the front end (csrc/host: image_io.cpp, jpeg_decode.cpp, gltf_loader.cpp + json.h) under
# AddressSanitizer + UBSan (host code only). Seeds: PNGs written by PIL (RGB, RGBA, palette, 16-bit grey, Adam7), JPEGs
# (baseline, progressive), and two .glb files written by pathtracer_amd.gltf_export (textured test scene with embedded
# PNGs, u16 and u32 indices). Everything is built and written under pathtracer_amd/build/ (git-ignored).
set -e
cd "$(dirname "$0")/.."
out=pathtracer_amd/build/loader_fuzz
mkdir -p $out
python3 - "$out" <<'PY'
import sys
import numpy as np
from PIL import Image
sys.path.insert(0, ".")
from pathtracer_amd import scenes, gltf_export
o = sys.argv[1]
y, x = np.mgrid[0:37, 0:53]
img = np.stack([128 + 100 * np.sin(x / 9.0) * np.cos(y / 7.0), 128 + 90 * np.cos(x / 5.0 + y / 11.0), (x * 2 + y) % 256, 255 - (x + y) % 200], -1)
img = (img + np.random.RandomState(1).normal(0, 20, img.shape)).clip(0, 255).astype(np.uint8)
Image.fromarray(img[..., :3]).save(f"{o}/rgb.png")
Image.fromarray(img).save(f"{o}/rgba.png")
Image.fromarray(img[..., :3]).convert("P", palette=Image.ADAPTIVE, colors=17).save(f"{o}/pal.png")
Image.fromarray((img[..., 0].astype(np.uint16) * 257)).save(f"{o}/grey16.png")
Image.fromarray(img[..., :3]).save(f"{o}/prog.jpg", "JPEG", progressive=True, subsampling=2)
Image.fromarray(img[..., :3]).save(f"{o}/base.jpg", "JPEG", subsampling=1, restart_marker_blocks=2)
s = scenes.textured_scene(grid=2)
s.images = [im[:16, :16].copy() for im in s.images]
open(f"{o}/textured_u16.glb", "wb").write(gltf_export.scene_to_glb(s, index_type="u16"))
open(f"{o}/cube_u32.glb", "wb").write(gltf_export.scene_to_glb(scenes.cube_scene(), index_type="u32"))
PY
H=pathtracer_amd/csrc/host
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -I $H -I include \
  scripts/micro/loader_fuzz.cpp $H/image_io.cpp $H/jpeg_decode.cpp $H/gltf_loader.cpp -lz -o $out/loader_fuzz
$out/loader_fuzz $out/scratch.glb ${1:-3000} $out/rgb.png $out/rgba.png $out/pal.png $out/grey16.png $out/prog.jpg $out/base.jpg \
  $out/textured_u16.glb $out/cube_u32.glb
