# the CPU suite against an AddressSanitizer + UndefinedBehaviorSanitizer build of the oracle (sanitizers run on the CPU build only; the build goes to a scratch directory)
set -e
D=${TMPDIR:-/tmp}/oracle_asan
mkdir -p $D $D/../include
cp oracle/*.cpp oracle/*.hpp oracle/*.h oracle/Makefile $D/
cp include/vilfusion.h $D/../include/
make -s -C $D -j8 CXXFLAGS="-O1 -g -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer"
VILO_SO=$D/liboracle_vilf.so LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests -q -x -m "not gpu" "$@"
