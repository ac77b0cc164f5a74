#!/bin/bash
# tools/lab/build_variant.sh <name> <file.hip> [-Dflags...]: a second build of libsegmentron_hip.so whose
# <file.hip> is compiled with extra flags (kernel A/B through tools/lab/dw_ab, op_time.py ...):
# tools/lab/cand_<name>.so (git-ignored, travels with gpurun)
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../../segmentron_amd/csrc"
make -j8 ARCH=gfx950 all > /dev/null
OBJ=/tmp/variant_${NAME}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" -c $SRC -o $OBJ
OTHERS=$(ls *.o | grep -v strict | grep -v "^$(basename $SRC .hip).o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $OBJ -o ../../tools/lab/cand_${NAME}.so
echo built tools/lab/cand_${NAME}.so
