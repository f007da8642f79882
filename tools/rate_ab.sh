#!/bin/bash
# Same-box step rates of library variants: tools/rate_ab.sh "n1 n2 ..." name1 name2 ...   ("tree" = the tree's own library)
sizes=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = tree ]; then unset LJMD_LIBRARY; else export LJMD_LIBRARY=$R/variants/libljmd_$v.so; fi
    echo "== $v"; python3 $R/tools/n_sweep_rate.py $sizes 2>&1 | tail -n 12
  done
done
