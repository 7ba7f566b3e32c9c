#!/bin/bash
# Builds libmwengine.so for gfx950 in-tree (no GPU needed: hipcc cross-compiles).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
$HIPCC $FLAGS -shared mw_engine.hip mw_setup.hip mw_raster.hip mw_raster_mesh.hip mw_reset.hip mw_visible.hip mw_setup_pcg.hip mw_reset_pcg.hip mw_setup_sort.hip mw_setup_sort_pcg.hip -o libmwengine.so "$@"
echo "built $(pwd)/libmwengine.so"
