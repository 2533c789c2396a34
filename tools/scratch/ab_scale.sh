D=$GRAFT_REPO_ROOT/sfm-python_amd/sfmba
for r in 1 2 3; do for L in libsfmba_base.so libsfmba_new.so; do for c in cfg4 cfg3 cfg2; do echo -n "$L "; SFMBA_LIB=$D/$L python3 tools/solve_loop.py $c 40; done; done; done
for L in libsfmba_base.so libsfmba_new.so; do echo -n "$L K1 events "; SFMBA_K1_EVENTS=1 SFMBA_LIB=$D/$L python3 tools/solve_loop.py cfg4 40; done
for L in libsfmba_base.so libsfmba_new.so; do echo -n "$L "; SFMBA_LIB=$D/$L python3 tools/solve_loop.py cfg5 6 32; done
SFMBA_LIB=$D/libsfmba_new.so SFMBA_DEBUG=trace_timing=1 python3 tools/solve_loop.py cfg4 3 2>&1 | grep "sfmba: solve\|upload_x" | tail -4
SFMBA_LIB=$D/libsfmba_new.so python3 bench.py --no-cpu-baseline --no-per-call | tail -1 | cut -c1-400
