mkdir -p gpurun_out
for r in 1 2; do
for v in 0 1; do
  GCM_ONE_WAVE=$v timeout -k 10 300 python bench.py --config cfg2 --no-cpu-baseline --headline-only > gpurun_out/b.log 2>&1
  tail -1 gpurun_out/b.log | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('one_wave=$v',d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['kernel_ms'])"
done; done
