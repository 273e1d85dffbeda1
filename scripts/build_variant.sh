#!/bin/bash
# Build libpathtracer_amd.so of a git revision (default: the working tree) into variants/<name>.so
# for same-box A/B runs (scripts/ab.sh). Usage: scripts/build_variant.sh <name> [git-rev] [extra hipcc flags...]
set -e
cd "$(dirname "$0")/.."
name=$1; rev=${2:-WORKTREE}; shift; shift || true
mkdir -p variants
if [ "$rev" = "WORKTREE" ]; then
  PT_LIB_OUT=$PWD/variants/$name.so PT_EXTRA_HIPCC_FLAGS="$*" python3 -c "from pathtracer_amd import build; build.build(force=True)"
else
  tmp=$(mktemp -d)
  git archive "$rev" | tar -x -C "$tmp"
  (cd "$tmp" && rm -f pathtracer_amd/csrc/host/main.cpp && PT_LIB_OUT=$OLDPWD/variants/$name.so PT_EXTRA_HIPCC_FLAGS="$*" python3 -c "from pathtracer_amd import build; build.build(force=True)")
  rm -rf "$tmp"
fi
ls -la variants/$name.so
