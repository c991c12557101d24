#!/bin/bash
# build a variant of the working tree with extra flags into tools/bin/<name>
set -e
name=$1; shift
root=/root/repo; tmp=$(mktemp -d); out=$root/tools/bin/$name; mkdir -p $out
objs=""
for f in $root/nfai_amd/csrc/*.hip; do o=$tmp/$(basename $f .hip).o; hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -ffp-contract=off -fno-fast-math -w "$@" -c $f -o $o & objs="$objs $o"; done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libnfai_hip.so $objs; rm -rf $tmp; echo $out/libnfai_hip.so
