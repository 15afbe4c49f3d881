mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_host.py -q -m gpu -x -k "wire or host or calibrated or c_abi or scheduler" > gpurun_out/r02/pytest_wire.log 2>&1; tail -4 gpurun_out/r02/pytest_wire.log
bash tools/profile_round.sh r02 2>&1 | tail -3
python tools/configs.py gpurun_out/r02_configs.json > gpurun_out/r02_configs.log 2>&1; python - <<PY
import json
for r in json.load(open('gpurun_out/r02_configs.json')):
  print(r['config'][:70].ljust(72), 'ms %.3f' % r['ms'], 'launches', r['launches'], r.get('schedule'), r.get('pass_us'))
PY
