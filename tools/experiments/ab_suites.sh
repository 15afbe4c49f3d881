# the backend's A/B switches must keep working: parity subsets under each
set -o pipefail
sel="corpus_2d or corpus_3d or golden or integer_window or temporal or hand_written"
for env in "SODA_HIP_WINDOWS=0" "SODA_HIP_INLINE=0" "SODA_HIP_SLIDE=0" "SODA_HIP_NO_CALIBRATE=1" "SODA_HIP_NO_PROBE=1"; do
  echo "== $env"
  env $env python -m pytest tests/test_hip_parity.py -x -q -k "$sel" 2>&1 | tail -1
done
echo "== SODA_HIP_SPLIT=side"
SODA_HIP_SPLIT=side python -m pytest tests/test_group.py tests/test_dist.py -x -q -m gpu -k "virtual_slabs or hidden or many_short or chained" 2>&1 | tail -1
