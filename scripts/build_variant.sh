#!/bin/bash
# Diagnostic: builds exp/liblacx_<name>.so from the current sources with extra -D flags on the kernels (scripts/kexp.py
# times such variants side by side).  usage: build_variant.sh <name> [flags...]
set -e
cd "$(dirname "$0")/../lossless-audio-codec_amd"
NAME=$1; shift
mkdir -p ../exp build
make -s liblacx.so >/dev/null
hipcc -O3 -std=c++20 -fPIC -Wall -Wno-unused-function "$@" --offload-arch=gfx950 -Icsrc -I../include -I/opt/rocm/include -c csrc/kernels.hip -o build/kernels_$NAME.o
hipcc -shared -o ../exp/liblacx_$NAME.so build/kernels_$NAME.o build/decode.o build/wide.o build/emit.o build/lacx_api.o -lpthread
echo built exp/liblacx_$NAME.so
