#!/bin/bash
# Build libpfd_hip.so of a git revision into variants/<name>.so (git-ignored; travels with gpurun) for same-box A/B runs:
#   tools/build_variant.sh <git-rev> <name>       then on the GPU box: tools/ab_bench.sh <name> ...
set -eu
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/pfd_variant_$NAME
rm -rf $W; git -C $ROOT worktree prune; git -C $ROOT worktree add --detach $W $REV > /dev/null
make -C "$W/prompt-free-diffusion_amd/csrc" -j8 ARCH=gfx950 ../libpfd_hip.so > /tmp/pfd_variant_$NAME.log 2>&1
mkdir -p $ROOT/variants
cp "$W/prompt-free-diffusion_amd/libpfd_hip.so" $ROOT/variants/$NAME.so
git -C $ROOT worktree remove --force $W
ls -la $ROOT/variants/$NAME.so
