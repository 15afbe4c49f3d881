#!/bin/bash
# The JIT cache (soda_amd/_jit_cache, content-addressed: compiler version +
# options + kernel text -> code object; git-ignored, travels to the GPU box
# with the tree like libsoda_hip.so) warmed with everything the GPU test suite
# compiles.  hiprtc takes 1-3 s per module and the suite builds ~1800 of them:
# 516 s cold, 135 s warm on one MI355X box (profiles/r05_suite_cold.log,
# r05_suite_warm.log) -- the driver's GPU run has a time limit.  Stale entries
# are harmless (never looked up); any edit to soda_rt.h or a generator changes
# the keys of what it touches, those modules are simply compiled again.
#
#   on the GPU box (gpurun):  tools/warm_jit_cache.sh collect      (or collect-new)
#   here, afterwards:         tools/warm_jit_cache.sh install      (or install-new)
set -o pipefail
cd "$(dirname "$0")/.."
case "$1" in
  collect)
    # (the FULL random-program seed sets: tests/fuzz.py then finds them in the
    # cache and runs them by default; outside gpurun_out while it fills, one
    # compressed file back -- gpurun merges at most 64 MiB)
    export SODA_HIP_CACHE=/tmp/soda_jit_cache_$$
    rm -rf "$SODA_HIP_CACHE"; mkdir -p "$SODA_HIP_CACHE"
    (time python -m pytest tests -m gpu -q --fuzz-budget 2) > gpurun_out/jit_cache_suite.log 2>&1
    tail -4 gpurun_out/jit_cache_suite.log
    du -sh "$SODA_HIP_CACHE"; ls "$SODA_HIP_CACHE" | wc -l
    rm -rf gpurun_out/jit_cache
    tar -C "$SODA_HIP_CACHE" -czf gpurun_out/jit_cache.tgz . && ls -la gpurun_out/jit_cache.tgz
    rm -rf "$SODA_HIP_CACHE"
    ;;
  collect-new)
    # the same on top of the cache the tree already carries: only the modules
    # it does not hold yet are compiled and come back (after a change that
    # touches few generators)
    export SODA_HIP_CACHE=/tmp/soda_jit_cache_$$
    rm -rf "$SODA_HIP_CACHE"; mkdir -p "$SODA_HIP_CACHE"
    cp -r soda_amd/_jit_cache/. "$SODA_HIP_CACHE"/ 2>/dev/null
    touch /tmp/soda_jit_marker_$$
    (time python -m pytest tests -m gpu -q --fuzz-budget 2) > gpurun_out/jit_cache_suite.log 2>&1
    tail -4 gpurun_out/jit_cache_suite.log
    (cd "$SODA_HIP_CACHE" && find . -type f -newer /tmp/soda_jit_marker_$$ > /tmp/soda_jit_new_$$ &&
     wc -l < /tmp/soda_jit_new_$$ && tar -czf "$OLDPWD/gpurun_out/jit_cache_new.tgz" -T /tmp/soda_jit_new_$$)
    ls -la gpurun_out/jit_cache_new.tgz
    rm -rf "$SODA_HIP_CACHE" /tmp/soda_jit_marker_$$ /tmp/soda_jit_new_$$
    ;;
  install-new)
    [ -f gpurun_out/jit_cache_new.tgz ] || { echo "run 'collect-new' through gpurun first"; exit 1; }
    mkdir -p soda_amd/_jit_cache
    tar -C soda_amd/_jit_cache -xzf gpurun_out/jit_cache_new.tgz
    du -sh soda_amd/_jit_cache; ls soda_amd/_jit_cache | wc -l
    ;;
  install)
    [ -f gpurun_out/jit_cache.tgz ] || { echo "run 'collect' through gpurun first"; exit 1; }
    rm -rf soda_amd/_jit_cache; mkdir -p soda_amd/_jit_cache
    tar -C soda_amd/_jit_cache -xzf gpurun_out/jit_cache.tgz
    python -c "import __graft_entry__ as g; g.build()"     # + the pre-JIT list
    du -sh soda_amd/_jit_cache; ls soda_amd/_jit_cache | wc -l
    ;;
  *) echo "usage: $0 collect|install|collect-new|install-new"; exit 2;;
esac
