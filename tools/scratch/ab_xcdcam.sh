cd $GRAFT_REPO_ROOT
for r in 1 2; do for V in 0 1; do
SFMBA_DEBUG=xcd_cam=$V python3 tools/time_kernels.py cfg5 2,8 32
echo -n "xcd_cam=$V "; SFMBA_DEBUG=xcd_cam=$V python3 tools/solve_loop.py cfg5 6 32
done; done
for V in 0 1; do SFMBA_DEBUG=xcd_cam=$V python3 tools/time_kernels.py cfg5 2,8 64; echo -n "xcd_cam=$V "; SFMBA_DEBUG=xcd_cam=$V python3 tools/solve_loop.py cfg5 6 64; done
for V in 0 1; do SFMBA_DEBUG=xcd_cam=$V,xcd_chunks=1 python3 tools/time_kernels.py cfg4 2,5,8 64; echo -n "cfg4 xcd_chunks=1 xcd_cam=$V "; SFMBA_DEBUG=xcd_cam=$V,xcd_chunks=1 python3 tools/solve_loop.py cfg4 32 64; done
