#!/bin/bash
# build container: the ROUND-5 step kernel (one cw_step_fused_kernel, `paint` a runtime argument) with today's host side, as a throw-away library for
# tools/microbench/r06_step_variants.sh:  gym_craftingworld_amd/libcw_exp_r05step.so  (git-ignored; travels to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/../.."
T=$(mktemp -d)
cp gym_craftingworld_amd/csrc/{cw_engine.cpp,cw_layout.h,cw_mt.h} $T/
git show f5ad428:gym_craftingworld_amd/csrc/cw_kernels.hip > $T/cw_kernels.hip
sed -i "s#../../include/craftingworld.h#$PWD/include/craftingworld.h#" $T/cw_engine.cpp
(cd $T && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -shared -o $OLDPWD/gym_craftingworld_amd/libcw_exp_r05step.so -x hip cw_kernels.hip cw_engine.cpp)
rm -rf $T
