for flags in "-O1" "-mllvm -amdgpu-dpp-combine=false" "-mllvm -amdgpu-dpp-combine=false -mllvm -amdgpu-sdwa-peephole=0" "-O2" "-fno-slp-vectorize -mllvm -amdgpu-sdwa-peephole=0 -O1"; do
  echo "== $flags"
  SODA_HIP_EXTRA_FLAGS="$flags" python tools/experiments/r04_repro159b.py 2>&1 | grep "(520, 291) {}\|(300, 80) {'chunk"
done
