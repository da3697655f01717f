mkdir -p gpurun_out
# A/B of cfg2's kernels: GCM_ONE_WAVE=1 the one-wave step kernel, GCM_BPTT_PER_ITEM=1 the per-item backward
for r in 1 2; do
for v in "0 0" "1 1"; do
  set -- $v
  GCM_ONE_WAVE=$1 GCM_BPTT_PER_ITEM=$2 timeout -k 10 300 python bench.py --config cfg2 --no-cpu-baseline --headline-only > gpurun_out/b.log 2>&1
  tail -1 gpurun_out/b.log | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('one_wave=$1 per_item=$2',d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],{k.split('::')[-1][:30]:v for k,v in d['kernel_ms'].items()})"
done; done
