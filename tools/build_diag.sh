#!/bin/sh
# Diagnostic build of the library with in-kernel phase stamps (-DAZ_STAMP): alphazero_amd/libazamd_diag.so,
# selected at run time with AZAMD_LIB=alphazero_amd/libazamd_diag.so. Needs an up-to-date normal build first.
set -e
cd "$(dirname "$0")/../alphazero_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -DAZ_STAMP -c az_net_mfma.hip -o /tmp/az_net_mfma_stamp.o
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libazamd_diag.so az_engine.o az_net.o /tmp/az_net_mfma_stamp.o az_net_layers.o az_replay.o az_hostrng.o
