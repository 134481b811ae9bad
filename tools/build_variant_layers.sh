#!/bin/sh
# Second build of the library with extra defines for the layer-path kernels, for same-box A/B runs:
# tools/build_variant_layers.sh <name> -DX=1 ... -> alphazero_amd/libazamd_<name>.so (run with AZAMD_LIB=<that path>).
# Needs an up-to-date normal build first (the other objects are reused).
set -e
name=$1; shift
cd "$(dirname "$0")/../alphazero_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -save-temps=obj -c az_net_layers.hip -o /tmp/az_net_layers_$name.o
grep -E "^; (NumVgprs|ScratchSize)|^\s+\.amdhsa_kernel " /tmp/az_net_layers-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - | grep duo | sed 's/.*k_conv3_duo/k_conv3_duo/'
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libazamd_$name.so az_engine.o az_net.o az_net_mfma.o /tmp/az_net_layers_$name.o az_replay.o az_hostrng.o
