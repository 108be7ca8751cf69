#!/bin/bash
# tuning variants of the HIP engine for A/B runs: tools/build_variant.sh <name> <hipcc flags...>
#   -> maniac_mc_amd/variants/libmaniac_hip_<name>.so   (select with MANIAC_HIP_LIB=<path> in tools/bench_kernels.py)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p maniac_mc_amd/variants
hipcc --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -fopenmp "$@" -o maniac_mc_amd/variants/libmaniac_hip_$name.so \
    maniac_mc_amd/csrc/mgpu_engine.hip maniac_mc_amd/csrc/mgpu_launch.hip maniac_mc_amd/csrc/mgpu_lanes.hip maniac_mc_amd/csrc/mgpu_windows.hip \
    maniac_mc_amd/csrc/mgpu_host_setup.cpp maniac_mc_amd/csrc/mgpu_comm.cpp -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built maniac_mc_amd/variants/libmaniac_hip_$name.so
