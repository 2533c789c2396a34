# quick A/B of the committed build (HEAD~0 lib saved as libsfmba_prev.so when present) and the working build
D=$GRAFT_REPO_ROOT/sfm-python_amd/sfmba
for r in 1 2; do for L in libsfmba_prev.so libsfmba.so; do
[ -f $D/$L ] || continue
SFMBA_LIB=$D/$L python3 tools/time_kernels.py cfg4 2,5,8 64
echo -n "$L "; SFMBA_LIB=$D/$L python3 tools/solve_loop.py cfg4 64
echo -n "$L "; SFMBA_LIB=$D/$L python3 tools/solve_loop.py cfg3 64
done; done
for L in libsfmba_prev.so libsfmba.so; do [ -f $D/$L ] || continue; SFMBA_LIB=$D/$L python3 tools/time_kernels.py cfg5 2,5,8 32; echo -n "$L "; SFMBA_LIB=$D/$L python3 tools/solve_loop.py cfg5 6 32; done
