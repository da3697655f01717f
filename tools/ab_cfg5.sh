mkdir -p gpurun_out
for r in 1 2; do
for v in 0 1; do
  GCM_LEARNED_FOUR_WAVES=$v timeout -k 10 300 python bench.py --config cfg5 --no-cpu-baseline --headline-only > gpurun_out/b.log 2>&1
  tail -1 gpurun_out/b.log | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('four_waves=$v',d['value'],d['ms_per_step'],{k:v for k,v in d['kernel_ms'].items() if 'select' in k})"
done; done
