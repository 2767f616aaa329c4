#!/bin/bash
# Build a variant of libpmce_hip.so from the CURRENT tree with extra compile flags (A/B switches are -D macros that exist only while an
# experiment runs) into pmce_amd/variants/libpmce_hip_<tag>.so, for A/B runs inside ONE gpurun call (boxes differ by several per cent):
#   bash scripts/build_ab.sh <tag> "<flags>"      then      PMCE_LIB_PATH=pmce_amd/variants/libpmce_hip_<tag>.so python ...
set -eu
tag=$1; flags=${2:-}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/pmce_variant.XXXXXX)
mkdir -p "$tmp/pmce_amd" "$tmp/include"
cp "$root"/include/*.h "$tmp/include/"
cp "$root"/pmce_amd/*.py "$tmp/pmce_amd/"
mkdir -p "$tmp/pmce_amd/csrc"
cp "$root"/pmce_amd/csrc/*.hip "$root"/pmce_amd/csrc/*.cpp "$root"/pmce_amd/csrc/*.hpp "$tmp/pmce_amd/csrc/"
(cd "$tmp" && PMCE_EXTRA_HIPCC_FLAGS="$flags" python -c "import sys; sys.path.insert(0, '.'); import importlib.util as u; s = u.spec_from_file_location('b', 'pmce_amd/build.py'); b = u.module_from_spec(s); s.loader.exec_module(b); print(b.build(force=True))")
mkdir -p "$root/pmce_amd/variants"
cp "$tmp/pmce_amd/libpmce_hip.so" "$root/pmce_amd/variants/libpmce_hip_$tag.so"
rm -rf "$tmp"
echo "$root/pmce_amd/variants/libpmce_hip_$tag.so"
