# same-box A/B of the batched window solve: round-4 library, current two-kernel sequence, current k_iter. usage: bash tools/dev_ab_solve.sh [B]
B=${1:-4096}
VILF_SO=$PWD/tools/ab/r04.so python tools/dev_solve_time.py $B "r04 library"
python tools/dev_solve_time.py $B "current, two kernels"
VILF_FUSED=1 python tools/dev_solve_time.py $B "current, k_iter"
VILF_FUSED=1 VILF_NO_SLOTS=1 python tools/dev_solve_time.py $B "current, k_iter, no slots"
