#!/bin/bash
# Dev aid: a variant build of libppt_hip.so with extra compiler flags, for A/B runs on one box through PPT_HIP_LIB:
#   bash tools/build_variant.sh nt -DPPT_NT_STORES    ->  tools/_build/libppt_nt.so
#   python tools/ab_env.py C2 3 "PPT_HIP_LIB=" "PPT_HIP_LIB=$PWD/tools/_build/libppt_nt.so"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/tools/_build/var_$NAME
mkdir -p "$OUT"
for f in "$ROOT"/ppt_amd/csrc/*.hip; do
    b=$(basename "$f" .hip)
    extra=""
    case $b in fps|knn_group) extra="-ffp-contract=off";; esac
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -fno-gpu-rdc -fno-slp-vectorize $extra "$@" -c "$f" -o "$OUT/$b.o" &
    if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$OUT"/*.o -o "$ROOT/tools/_build/libppt_$NAME.so"
ls -la "$ROOT/tools/_build/libppt_$NAME.so"
