#!/bin/bash
# Dev helper: builds a copy of libarseg_hip.so with creff_rr.hip compiled under extra flags (e.g. -DRR_TIMING, -DRR_ABLATE=..).
#   bash tools/build_rr_variant.sh <name> [flags...]   ->  scratch/rr_libs/lib_<name>.so   (scratch/ is not tracked)
set -e
name=$1; shift
SRC=${SRC:-creff_rr}       # SRC=<file stem> builds another source of csrc/ under the flags instead
R=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$R/ar-seg_amd/lib/obj
mkdir -p /tmp/rr $R/scratch/rr_libs
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include "$@" -c $R/ar-seg_amd/csrc/$SRC.hip -o /tmp/rr/${SRC}_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC $(ls $OBJ/*.o | grep -v /$SRC.o) /tmp/rr/${SRC}_$name.o -o $R/scratch/rr_libs/lib_$name.so
