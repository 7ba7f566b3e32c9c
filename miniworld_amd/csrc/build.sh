#!/bin/bash
# Builds libmwengine.so for gfx950 in-tree (no GPU needed: hipcc cross-compiles).
set -e
cd "$(dirname "$0")"
make -j"$(nproc)" "$@" >/dev/null
echo "built $(pwd)/libmwengine.so"
