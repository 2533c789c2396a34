#!/bin/bash
# Untraced A/B of one debug option on back-to-back solves, alternating:  bash tools/ab_option.sh <workload> <bits> <option> [solves]
W=$1; B=$2; OPT=$3; N=${4:-20}
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for V in 0 1; do echo -n "$OPT=$V  "; SFMBA_DEBUG=$OPT=$V python3 tools/solve_loop.py $W $N $B; done; done
