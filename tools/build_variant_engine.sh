#!/bin/sh
# Like build_variant.sh, for the search / rules kernels: tools/build_variant_engine.sh <name> -DX=1 ...
set -e
name=$1; shift
cd "$(dirname "$0")/../alphazero_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c az_engine.hip -o /tmp/az_engine_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libazamd_$name.so /tmp/az_engine_$name.o az_net.o az_net_mfma.o az_net_layers.o az_replay.o az_hostrng.o
