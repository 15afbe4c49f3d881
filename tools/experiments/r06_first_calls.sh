#!/bin/bash
# First GPU calls of the next round: sustained A/B (tools/ab.py, ~2 s of GPU
# each) of shapes that round 5's last hour pointed at but had no minutes left
# to time.  Register counts from hiprtc here (no GPU):
#   xcorr    8192^2: default V8/P2 239 VGPRs (2 waves per SIMD); V4/P2 126,
#                    V4/P4 138 (3-4 waves) -- round 3 timed V4 at 160 registers
#   erosion  8192^2: default V8/P2 243; V4/P2 123
#   denoise3d 512^3: default V1 113; vec 2 + xshare 178 -- 539-553 against
#                    517-519 us in short windows (profiles/r05_denoise_shapes.jsonl)
# Run `bash tools/experiments/r06_first_calls.sh --compile-only` here first (the
# JIT cache travels), then the same through gpurun without the flag:
#   gpurun --timeout 120 -- 'bash tools/experiments/r06_first_calls.sh > gpurun_out/r06_ab.jsonl'
cd "$(dirname "$0")/../.."
flag=$1
ab() { python tools/ab.py "$@" $flag || echo "{\"error\": \"$1\"}"; }
ab xcorr.soda 8192 8192 --arm '{}' --arm '{"vec": 4}' --arm '{"vec": 4, "prefetch": 4}' --arm '{"vec": 4, "prefetch": 1}'
ab erosion.soda 8192 8192 --arm '{}' --arm '{"vec": 4}' --arm '{"vec": 4, "prefetch": 1}'
ab denoise3d.soda 512 512 512 --arm '{}' --arm '{"vec": 2, "xshare": true, "reg_budget": 1048576}' --arm '{"xshare": true}'
ab denoise2d.soda 8192 8192 --arm '{}' --arm '{"vec": 4, "prefetch": 8}' --arm '{"vec": 2, "prefetch": 2}'
ab sobel2d.soda 8192 8192 --arm '{}' --arm '{"vec": 4}'
ab blur.soda 16384 16384 --arm '{}' --arm '{"vec": 4}' --arm '{"prefetch": 4}'
ab heat3d.soda 512 512 512 --iterate 1 --arm '{}' --arm '{"vec": 2}'
# (not an A/B: round 5's last fix changed kStencilDistance for windows that lie
# ahead of the cell -- core.Stencil.stencil_distance now follows reference
# core.py:620-625 -- which widens the dense-view criterion for such programs;
# the GPU suite's wire cases have symmetric windows, so scan random ones again:)
if [ -z "$flag" ]; then
  timeout -k 10 300 python tools/fuzz_scan.py wire 0 400 | tail -3
fi
