#!/bin/sh
# Second build of the library with extra defines for same-box A/B runs: tools/build_variant.sh <name> -DX=1 ...
# -> alphazero_amd/libazamd_<name>.so, selected at run time with AZAMD_LIB=alphazero_amd/libazamd_<name>.so.
# Needs an up-to-date normal build first (the other objects are reused).
set -e
name=$1; shift
cd "$(dirname "$0")/../alphazero_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c az_net_mfma.hip -o /tmp/az_net_mfma_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libazamd_$name.so az_engine.o az_net.o /tmp/az_net_mfma_$name.o az_net_layers.o az_replay.o az_hostrng.o
