#!/bin/bash
# Build everything that travels to the GPU box (product library + oracle), then hand the command to gpurun.
# Usage: scripts/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python3 -m pathtracer_amd.build > /dev/null
make -C oracle > /dev/null
exec /usr/local/graft/bin/gpurun "$@"
