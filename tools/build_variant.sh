#!/bin/bash
# Builds another libljmd.so beside the tree's own for same-box A/B runs: tools/build_variant.sh NAME "-DFLAG=1 ..."
# -> variants/libljmd_NAME.so (git-ignored; travels to the GPU box).  LJMD_VARIANT_REV=<git rev> builds that revision's
# sources instead of the working tree's.  Measurement tool.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/variants
SRC=$R
if [ -n "$LJMD_VARIANT_REV" ]; then
  SRC=/tmp/ljmd_rev_$name; rm -rf $SRC; mkdir -p $SRC
  git -C $R archive "$LJMD_VARIANT_REV" include molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd/csrc | tar -x -C $SRC
fi
make -s -C "$SRC/molecular-dynamics-simulation---lennard-jones-monoatomic-fluid_amd/csrc" OBJDIR=/tmp/ljmd_obj_$name OUT=$R/variants/libljmd_$name.so EXTRA="$*"
echo "built variants/libljmd_$name.so ($* ${LJMD_VARIANT_REV:+rev $LJMD_VARIANT_REV})"
