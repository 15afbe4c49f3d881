#!/bin/bash
# Sanitizer runs of everything that is plain host code (CPU, no GPU needed):
#   1. libsoda_hip built with -fsanitize=address,undefined (make -C
#      soda_amd/csrc asan) under the tests that drive its pure entries --
#      launch geometry, schedules, launch planning by brute force, group
#      planning, the pack / unpack threads of the host-array entry;
#   2. the same sources under -fsanitize=thread for the worker pool;
#   3. the CPU oracle's GENERATED loop nests under ASan + UBSan (corpus, fuzz
#      seeds, independent nests): an out-of-box load in the oracle is caught.
# The reference builds its csim with -fsanitize=address
# (reference tests/test-cpp-host.sh:23, tests/test-cluster.sh:35).
# usage: tools/sanitize.sh [report file]      (default profiles/r05_sanitizers.txt)
set -o pipefail
cd "$(dirname "$0")/.."
out=${1:-profiles/r05_sanitizers.txt}
# (libstdc++ preloaded next to the sanitizer runtime: the runtime's __cxa_throw
# interceptor looks its real function up when it starts, and in a Python that
# loads libstdc++ later, with an extension module, finds none -- the first C++
# exception in ANY extension then ends the process with "CHECK failed ...
# real___cxa_throw != 0".  SciPy's HiGHS, which solves the produce-offset ILP,
# throws one whenever its interior-point thread is interrupted: now and then.)
stdcxx=$(gcc -print-file-name=libstdc++.so.6)
asan="$(gcc -print-file-name=libasan.so) $stdcxx"
tsan="$(gcc -print-file-name=libtsan.so) $stdcxx"
lib=$PWD/soda_amd/_sanitize
fail=0
{
  echo "# tools/sanitize.sh, $(date -u +%FT%TZ), $(gcc --version | head -1)"
  make -C soda_amd/csrc asan tsan 2>&1 | tail -3 || fail=1
  # the product library up to date BEFORE a sanitizer runtime is preloaded (the
  # tests' `built` fixture would otherwise run hipcc under it)
  python -c "import __graft_entry__ as g; g.build_library()" || fail=1
  # (SODA_HIP_NO_TORCH: the library binds to /opt/rocm's HIP runtime instead of
  # PyTorch's copy -- nothing here touches a GPU, and PyTorch under a preloaded
  # sanitizer runtime is slow and noisy)
  run() {   # label, preload, library, extra env..., -- pytest args
    local label=$1 preload=$2 which=$3; shift 3
    echo; echo "## $label"
    env SODA_HIP_NO_TORCH=1 LD_PRELOAD="$preload" SODA_HIP_LIBRARY=$which \
        ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 \
        UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
        TSAN_OPTIONS=halt_on_error=1:report_signal_unsafe=0 \
        "$@" > /tmp/soda_sanitize_section.log 2>&1
    local rc=$?
    tail -6 /tmp/soda_sanitize_section.log
    [ $rc -eq 0 ] || cp /tmp/soda_sanitize_section.log "/tmp/soda_sanitize_failed_$(date +%s).log"
    echo "exit code $rc"
    [ $rc -eq 0 ] || fail=1
  }
  run "libsoda_hip, ASan + UBSan: geometry, schedules, launch planning, group planning, pack / unpack" \
      "$asan" $lib/libsoda_hip_asan.so \
      python -m pytest -q -m "not gpu" -p no:cacheprovider \
      tests/test_codegen.py tests/test_group.py tests/test_host.py \
      tests/test_stream.py "tests/test_dist.py::test_exchange_interval_by_the_librarys_cost_choice" \
      --deselect tests/test_group.py::test_geometry_of_a_tall_stream_is_cheap
  # (deselected: a wall-clock assertion -- plan_geometry of a 2M-row stream in
  # under 10 ms -- that an instrumented build on a busy box misses now and then)
  run "libsoda_hip, TSan: the worker pool of the host-array entry (pack / unpack on 8 threads)" \
      "$tsan" $lib/libsoda_hip_tsan.so \
      python -m pytest -q -m "not gpu" -p no:cacheprovider tests/test_host.py -k "pack_and_unpack"
  run "generated oracle nests, ASan + UBSan: corpus, golden vectors, fuzz seeds, independent nests" \
      "$asan" $lib/libsoda_hip_asan.so SODA_ORACLE_SANITIZE=1 SODA_ORACLE_BUILD=/tmp/soda_oracle_asan_$$ \
      python -m pytest -q -m "not gpu" -p no:cacheprovider \
      tests/test_oracle.py tests/test_fuzz.py tests/test_fuzz_nest.py
  rm -rf /tmp/soda_oracle_asan_$$
  echo
  echo "# not run: TSan over SODA_HIP_GROUP_THREADS (one enqueueing thread per slab):"
  echo "# a group needs a GPU to exist (soda_hip_group_create loads code objects), and"
  echo "# sanitizer runtimes are not available on the GPU pool; its planning half"
  echo "# (soda_hip_group_plan, soda_hip_plan_launches) is pure and is covered above."
  echo "# overall: $([ $fail -eq 0 ] && echo clean || echo FAILURES)"
} | tee "$out"
exit $fail
