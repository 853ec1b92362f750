#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over everything of this repo that runs on the CPU and is compiled code (the GPU pool has no
# sanitizer): the C oracle (oracle/pqp_oracle.c) and the HOST BUILDS OF THE DEVICE ALGORITHM SOURCES (tests/emu/lq_emu.cpp = csrc/pqp_path_lq.hpp,
# tests/emu/lane_emu.cpp = csrc/pqp_path_lane.hpp + csrc/pqp_banded_qp.hpp) - an index that runs off a lane's arrays or a workspace block there does so in the
# kernels too.  Builds the three libraries with -fsanitize=address,undefined into ab/asan/ and runs the CPU tests that drive them under LD_PRELOAD=libasan.
#   tools/sanitize_cpu.sh [pytest args ...]            (default: the emulation and oracle test files)
set -e
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root"
out=ab/asan; mkdir -p $out
F="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -shared"
g++ $F -std=c++17 -o $out/liblq_emu.so tests/emu/lq_emu.cpp &
g++ $F -std=c++17 -DPQP_EMU_DIET=0 -o $out/liblane_emu.so tests/emu/lane_emu.cpp &
gcc $F -fopenmp -o $out/libpqp_oracle.so oracle/pqp_oracle.c -lm &
wait
tests=${@:-tests/test_lq_emulation.py tests/test_lane_emulation.py tests/test_banded_core_emulation.py tests/test_oracle_c.py tests/test_highs_pin.py}
PQP_SANITIZED_LIBS=$root/$out LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=8 \
  python -m pytest $tests -x -q -m "not gpu" -p no:cacheprovider
