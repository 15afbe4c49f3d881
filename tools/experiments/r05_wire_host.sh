#!/bin/bash
# <app>_kernel on HOST banks (what the reference host calls under
# SODA_CPP_BINDING): every tensor in place on the dense view -> the banded
# host-array entry, against whole banks through the staging rings
# (SODA_HIP_STREAM_NO_BANDS=1, rounds 1-5a).  -> gpurun_out/r05_wire_host.jsonl
set -e
out=gpurun_out/r05_wire_host.jsonl
mkdir -p gpurun_out
: > $out
g=tests/golden/soda
for nb in "" 1; do
  export SODA_HIP_STREAM_NO_BANDS=$nb
  [ -z "$nb" ] && unset SODA_HIP_STREAM_NO_BANDS
  echo "{\"SODA_HIP_STREAM_NO_BANDS\": \"$nb\"}" >> $out
  python tools/streambench.py --host --steps 3 --soda $g/blur.soda --tile 16384 --extent 16384 16384 | grep host_banks >> $out
  python tools/streambench.py --host --steps 3 --soda $g/jacobi2d.soda --tile 8192 --iterate 100 --extent 8192 8192 | grep host_banks >> $out
  python tools/streambench.py --host --steps 3 --soda $g/heat3d.soda --tile 512 512 --iterate 50 --extent 512 512 512 | grep host_banks >> $out
  python tools/streambench.py --host --steps 3 --soda $g/blur.soda --tile 16384 --banks 4 --extent 16384 16384 | grep host_banks >> $out
  python tools/streambench.py --host --steps 3 --soda $g/jacobi2d.soda --tile 8192 --banks 2 --iterate 100 --extent 8192 8192 | grep host_banks >> $out
done
cat $out
