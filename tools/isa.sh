#!/bin/bash
# Device ISA of the engine library for gfx950 (no GPU needed).  Usage: tools/isa.sh <out.s> [source root = repo]
OUT=${1:-/tmp/lzani_isa.s}; SRC=${2:-$(cd "$(dirname "$0")/.." && pwd)}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value --cuda-device-only -S -o "$OUT" "$SRC/lz-ani_amd/csrc/lzani_hip.hip" || exit 1
python3 "$(dirname "$0")/isa_stats.py" "$OUT"
