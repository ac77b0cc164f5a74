#!/bin/bash
# builds tools/lab/gemm_lab against the production objects (+ a -DGL_LAB build of the glds kernel)
set -e
cd "$(dirname "$0")/../.."
make -s -C segmentron_amd/csrc -j8
H=/opt/rocm/bin/hipcc
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off"
$H $F -DGL_LAB -c segmentron_amd/csrc/conv_gemm_glds.hip -o /tmp/conv_gemm_glds_lab.o
$H $F -c tools/lab/gemm_lab.hip -o /tmp/gemm_lab.o
$H --offload-arch=gfx950 /tmp/gemm_lab.o segmentron_amd/csrc/core.o segmentron_amd/csrc/conv_gemm_px256.o /tmp/conv_gemm_glds_lab.o -o tools/lab/gemm_lab
echo built tools/lab/gemm_lab
$H $F -c tools/lab/wgrad_lab.hip -o /tmp/wgrad_lab.o
$H --offload-arch=gfx950 /tmp/wgrad_lab.o segmentron_amd/csrc/core.o segmentron_amd/csrc/conv_gemm_wgrad.o segmentron_amd/csrc/conv_gemm_wgrad_glds.o segmentron_amd/csrc/conv3x3_direct.o -o tools/lab/wgrad_lab
echo built tools/lab/wgrad_lab
