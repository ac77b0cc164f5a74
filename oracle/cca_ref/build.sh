#!/bin/sh
# TEST INFRASTRUCTURE: builds oracle/_ref/libcca_ref.so — the reference's criss-cross-attention
# CUDA kernels (ca_cuda.cu:8-177) compiled as host C++ — from the reference checkout where it
# lies.  Outputs only into oracle/_ref/ (git-ignored; it travels to the GPU box like any built
# .so).  Not the reference's build system: one awk + one g++.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${SEGMENTRON_REFERENCE:-/root/reference}/segmentron/modules/csrc/criss_cross_attention/ca_cuda.cu"
OUT="$HERE/../_ref"
[ -f "$REF" ] || { echo "cca_ref: $REF not found" >&2; exit 3; }
mkdir -p "$OUT"
# the kernel templates: everything above the host wrappers, without the ATen / THC includes
awk '/^namespace segmentron/{exit} !/^#include/{print}' "$REF" > "$OUT/ca_kernels.inc"
g++ -O2 -std=c++14 -shared -fPIC -I "$OUT" "$HERE/driver.cpp" -o "$OUT/libcca_ref.so"
echo "built $OUT/libcca_ref.so"
