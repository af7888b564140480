# same-box A/B of the batched window solve: libraries tools/ab/<name>.so given as arguments (default: r04 prev), then the library in the tree. usage: bash tools/dev_ab_solve.sh [B] [names...]
B=${1:-4096}; shift
NAMES=${@:-r04 prev}
for n in $NAMES; do VILF_SO=$PWD/tools/ab/$n.so python tools/dev_solve_time.py $B "$n"; done
python tools/dev_solve_time.py $B "tree"
