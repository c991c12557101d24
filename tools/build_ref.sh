#!/bin/bash
# Build libnfai_hip.so from the sources of a git revision into tools/bin/<name>/ (A/B runs on the same GPU box:
#   NFAI_HIP_LIB=tools/bin/<name>/libnfai_hip.so python bench.py ...).  tools/bin/ is git-ignored but travels with gpurun.
#   tools/build_ref.sh <git-ref> <name> [extra hipcc flags]
set -e
ref=$1; name=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$ref" nfai_amd/csrc include | tar -x -C "$tmp"
out="$root/tools/bin/$name"; mkdir -p "$out"
objs=""
for f in "$tmp"/nfai_amd/csrc/*.hip; do
  o="$tmp/$(basename "$f" .hip).o"
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -ffp-contract=off -fno-fast-math -w "$@" -c "$f" -o "$o" &
  objs="$objs $o"
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o "$out/libnfai_hip.so" $objs
rm -rf "$tmp"
echo "$out/libnfai_hip.so"
