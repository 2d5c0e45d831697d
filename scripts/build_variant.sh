#!/bin/bash
# Diagnostic: builds exp/liblacx_<name>.so from the current sources with extra -D flags on the kernels (scripts/kexp.py
# times such variants side by side).  usage: build_variant.sh <name> [flags...]
set -e
cd "$(dirname "$0")/../lossless-audio-codec_amd"
NAME=$1; shift
mkdir -p ../exp build
make -s liblacx.so >/dev/null
for u in k_front k_analyze k_emit; do
  hipcc -O3 -std=c++20 -fPIC -Wall -Wno-unused-function "$@" --offload-arch=gfx950 -Icsrc -I../include -I/opt/rocm/include -c csrc/$u.hip -o build/${u}_$NAME.o &
done
wait
hipcc -shared -o ../exp/liblacx_$NAME.so build/k_front_$NAME.o build/k_analyze_$NAME.o build/k_emit_$NAME.o build/decode.o build/wide.o build/emit.o build/api_core.o build/api_pipeline.o build/api_encode.o build/api_decode.o build/api_fanout.o -lpthread -ldl
echo built exp/liblacx_$NAME.so
