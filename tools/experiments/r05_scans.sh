#!/bin/bash
# Random scans after round 5's changes (ABI 7, the host-array entry through the
# pinned ring and bands, group load / store through the rings): seeds beyond
# the suite's, every family, bit for bit against the C oracle.  Half of them
# with 24 KiB staging chunks, which puts bands and multi-chunk rings under
# every `Program.run` of the scan.
out=gpurun_out/r05_scans.txt
: > $out
run() {
  echo "## $*" >> $out
  "$@" 2>&1 | grep -v amdgpu.ids | tail -4 >> $out
}
run python tools/fuzz_scan.py generic 1200 1400
run env SODA_HIP_HOST_CHUNK_KB=24 python tools/fuzz_scan.py generic 1400 1600
run env SODA_HIP_HOST_CHUNK_KB=24 python tools/fuzz_scan.py rich 600 760
run env SODA_HIP_HOST_CHUNK_KB=24 python tools/fuzz_scan.py window 500 620
run python tools/fuzz_scan.py options 700 820
run env SODA_HIP_HOST_CHUNK_KB=24 python tools/fuzz_scan.py group 600 700
run env SODA_HIP_HOST_CHUNK_KB=24 python tools/fuzz_scan.py wire 500 600
run python tools/fuzz_scan.py deep 420 460
run python tools/fuzz_scan.py wide 600 680
cat $out
