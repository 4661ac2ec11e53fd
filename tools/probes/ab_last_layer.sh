set -x
timeout 1500 python -m pytest tests/test_gpu_trueshape.py tests/test_gpu_model.py tests/test_gpu_streams.py tests/test_gpu_multistream.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --multi-stream 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SPARSE stream300', d['value'], d['ms_per_step'])"
  MMDUET_FULL_LAST_LAYER=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --multi-stream 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FULL   stream300', d['value'], d['ms_per_step'])"
  python bench.py --config ground600 --steps 2 --warmup 1 --no-cpu-baseline --multi-stream 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SPARSE ground600', d['value'], d['ms_per_step'])"
  MMDUET_FULL_LAST_LAYER=1 python bench.py --config ground600 --steps 2 --warmup 1 --no-cpu-baseline --multi-stream 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FULL   ground600', d['value'], d['ms_per_step'])"
done
