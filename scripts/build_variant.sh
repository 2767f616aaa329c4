#!/bin/bash
# Build libpmce_hip.so of another git revision into pmce_amd/variants/libpmce_hip_<tag>.so, for A/B runs inside ONE gpurun call
# (boxes differ by several per cent, so variants are only comparable within a call):  PMCE_LIB_PATH=<that file> python bench.py ...
#   bash scripts/build_variant.sh <git-rev> <tag>
set -eu
rev=$1; tag=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/pmce_variant.XXXXXX)
git -C "$root" archive "$rev" pmce_amd include | tar -x -C "$tmp"
(cd "$tmp" && python -c "import pmce_amd.build as b; print(b.build())")
mkdir -p "$root/pmce_amd/variants"
cp "$tmp/pmce_amd/libpmce_hip.so" "$root/pmce_amd/variants/libpmce_hip_$tag.so"
rm -rf "$tmp"
echo "$root/pmce_amd/variants/libpmce_hip_$tag.so"
