#!/bin/bash
# A/B of library builds inside ONE gpurun call (boxes of the pool differ by 3-5 %): bash tools/ab_libs.sh out.txt lib1 lib2 ...
# per library and round: back-to-back kernel timings at cfg4 and cfg5 (fp32 storage), then solve loops.
OUT=$1; shift
D=$GRAFT_REPO_ROOT/sfm-python_amd/sfmba
: > $OUT
for round in 1 2; do
  for L in "$@"; do
    SFMBA_LIB=$D/$L python3 tools/time_kernels.py cfg4 0,7,2,4,5,8 64 >> $OUT 2>&1
    SFMBA_LIB=$D/$L python3 tools/time_kernels.py cfg5 0,7,2,4,5,8 32 >> $OUT 2>&1
  done
done
for L in "$@"; do
  for c in cfg4 cfg3 cfg2; do echo -n "$L " >> $OUT; SFMBA_LIB=$D/$L python3 tools/solve_loop.py $c >> $OUT 2>&1; done
  echo -n "$L " >> $OUT; SFMBA_LIB=$D/$L python3 tools/solve_loop.py cfg5 6 32 >> $OUT 2>&1
  echo -n "$L " >> $OUT; SFMBA_LIB=$D/$L python3 tools/solve_loop.py cfg5 6 64 >> $OUT 2>&1
done
cat $OUT
