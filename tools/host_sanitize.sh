#!/bin/bash
# Host-side code (presolve, tiled builder, transposes, shard extraction, MPS reader) under AddressSanitizer +
# UndefinedBehaviorSanitizer on the CPU (GPU ASan is not available on the test pool).  Needs hipcc for the headers
# only; nothing touches a GPU.    usage: tools/host_sanitize.sh
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; SRC="$ROOT/hpr-lp-c_amd/csrc"; OUT="${TMPDIR:-/tmp}/hprlp_asan"; mkdir -p "$OUT"
FLAGS="-O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -w -I$ROOT/include -I$SRC --offload-arch=gfx950 -x hip --cuda-host-only"
for f in presolve presolve_stages reorder tiled host_model mps_reader dist gen alloc; do /opt/rocm/bin/hipcc $FLAGS -c "$SRC/$f.cpp" -o "$OUT/$f.o"; done
/opt/rocm/bin/hipcc $FLAGS -c "$ROOT/tools/host_sanitize_driver.cpp" -o "$OUT/drv.o"
/opt/rocm/bin/hipcc -fsanitize=address,undefined -o "$OUT/drv" "$OUT"/{drv,presolve,presolve_stages,reorder,tiled,host_model,mps_reader,dist,gen,alloc}.o -lz -ldl
cd "$ROOT" && ASAN_OPTIONS=detect_leaks=1:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1 "$OUT/drv"
