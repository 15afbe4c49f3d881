mkdir -p gpurun_out/r02
bash tools/profile_round.sh r02 2>&1 | tail -2 | cut -c1-1500
python tools/configs.py gpurun_out/r02_configs.json > gpurun_out/r02_configs.log 2>&1; python - <<PY
import json
for r in json.load(open('gpurun_out/r02_configs.json')):
  print(r['config'][:70].ljust(72), 'ms %.3f' % r['ms'], 'launches', r['launches'], r.get('schedule'), r.get('pass_us'))
PY
bash tools/profile_configs.sh r02 2>&1 | tail -8
python tools/corpus_bench.py > gpurun_out/r02_corpus.jsonl 2>/dev/null; python tools/corpus_bench.py > gpurun_out/r02_corpus.jsonl 2>/dev/null; cut -c1-260 gpurun_out/r02_corpus.jsonl
python tools/model_check.py --fuse 12 8 4 --out gpurun_out/r02_model_check.jsonl 2>/dev/null | cut -c1-700
