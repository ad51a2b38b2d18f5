#!/bin/bash
# Alternative build of libwsnark.so with extra compile-time switches, for A/B runs (tools/ab_msm.py):
#   tools/build_variant.sh strict "-DWS_MADD_WIDE=0"      ->  tools/alt/libwsnark_strict.so
set -e
NAME=$1; FLAGS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B="$ROOT/tools/alt/build_$NAME"
mkdir -p "$B"
cd "$ROOT/wasmsnark_amd/csrc"
OBJS=""
for f in context ntt msm calch dist prove fixedbase synth selftest verify cabi; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c $f.hip -o "$B/$f.o" &
  OBJS="$OBJS $B/$f.o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/tools/alt/libwsnark_$NAME.so" $OBJS -lpthread
echo built "$ROOT/tools/alt/libwsnark_$NAME.so"
