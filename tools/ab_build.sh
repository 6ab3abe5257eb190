#!/bin/bash
# A/B builds of one translation unit: tools/ab_build.sh <unit> <name> "<-D flags>"  ->  build/ab/lib_<name>.so  (DORY_LIB_PATH selects it)
set -e
cd "$(dirname "$0")/.."
UNIT=$1; NAME=$2; FLAGS=$3
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value $FLAGS -c dorylus_amd/csrc/$UNIT.hip -o build/ab/${UNIT}_$NAME.o
OBJS=$(ls dorylus_amd/csrc/*.o | grep -v "/$UNIT.o"); HOST=$(ls dorylus_amd/host/*.o | grep -v _main)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp $OBJS build/ab/${UNIT}_$NAME.o $HOST -L/opt/rocm/lib -lrccl -o build/ab/lib_$NAME.so
echo build/ab/lib_$NAME.so
