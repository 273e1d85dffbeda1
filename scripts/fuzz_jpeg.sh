#!/bin/bash
# Damaged-stream fuzz of csrc/host/jpeg_decode.cpp under AddressSanitizer + UBSan (host code only). Seeds: four small
# streams written by PIL (progressive 4:2:0, progressive 4:4:4 low quality, progressive 4:2:2 with restart intervals,
# baseline 4:2:0). Everything is built and written under pathtracer_amd/build/ (git-ignored).
set -e
cd "$(dirname "$0")/.."
out=pathtracer_amd/build/jpeg_fuzz
mkdir -p $out
python3 - "$out" <<'PY'
import sys
import numpy as np
from PIL import Image
y, x = np.mgrid[0:67, 0:93]
img = np.stack([128 + 100 * np.sin(x / 9.0) * np.cos(y / 7.0), 128 + 90 * np.cos(x / 5.0 + y / 11.0), (x * 2 + y) % 256], -1)
img = (img + np.random.RandomState(1).normal(0, 20, img.shape)).clip(0, 255).astype(np.uint8)
for i, kw in enumerate([dict(progressive=True, subsampling=2), dict(progressive=True, subsampling=0, quality=30),
                        dict(progressive=True, subsampling=1, restart_marker_blocks=3), dict(subsampling=2)]):
    Image.fromarray(img).save(f"{sys.argv[1]}/s{i}.jpg", "JPEG", **kw)
PY
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -I pathtracer_amd/csrc/host \
  scripts/micro/jpeg_fuzz.cpp pathtracer_amd/csrc/host/jpeg_decode.cpp -o $out/jpeg_fuzz
$out/jpeg_fuzz $out/s0.jpg $out/s1.jpg $out/s2.jpg $out/s3.jpg
