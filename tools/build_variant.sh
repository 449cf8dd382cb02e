#!/bin/bash
# tools/build_variant.sh <name> [flags...] -- kyber_amd/variants/libkyber_hip_<name>.so: the library built with extra
# compiler flags (-D switches of experiments), for same-box A/B runs through KYBER_HIP_LIBRARY (tools/lib_ab.sh)
set -e
name=$1; shift
cd "$(dirname "$0")/../kyber_amd/csrc"
mkdir -p ../variants ../../build
C="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC"
$C "$@" -c kyber_hip.hip -o ../../build/kyber_hip_$name.o &
$C "$@" -fno-slp-vectorize -c kyber_arith.hip -o ../../build/kyber_arith_$name.o &
wait
$C -shared -o ../variants/libkyber_hip_$name.so ../../build/kyber_hip_$name.o ../../build/kyber_arith_$name.o
