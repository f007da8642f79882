#!/bin/bash
# Same-box timing of library variants (tools/build_variant.sh): tools/variant_probe.sh MODE N name1 name2 ...
# MODE = 0 fp64 | 1 mixed; prints probe_force.py's line per variant ("tree" = the tree's own library), twice round robin.
mode=$1; n=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = tree ]; then unset LJMD_LIBRARY; else export LJMD_LIBRARY=$R/variants/libljmd_$v.so; fi
    echo -n "$v: "; PROBE_MODE=$mode PROBE_N=$n PROBE_ROUNDS=1 PROBE_STEPS=${PROBE_STEPS:-8} python3 $R/tools/probe_force.py 2>&1 | tail -n 1
  done
done
