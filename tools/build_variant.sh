#!/bin/bash
# Builds another libljmd.so beside the tree's own for same-box A/B runs: tools/build_variant.sh NAME "-DFLAG=1 ..."
# -> variants/libljmd_NAME.so (git-ignored; travels to the GPU box).  Measurement tool.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/variants
make -s -C "$R/molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd/csrc" OBJDIR=/tmp/ljmd_obj_$name OUT=$R/variants/libljmd_$name.so EXTRA="$*"
echo "built variants/libljmd_$name.so ($*)"
