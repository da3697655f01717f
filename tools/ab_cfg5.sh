mkdir -p gpurun_out
# A/B of cfg5's kernels: GCM_LEARNED_FOUR_WAVES=1 the four-wave forward step, GCM_BPTT_MLP_BLOCKS=1 the 32-row-block pass B2
for r in 1 2; do
for v in "0 0" "1 1"; do
  set -- $v
  GCM_LEARNED_FOUR_WAVES=$1 GCM_BPTT_MLP_BLOCKS=$2 timeout -k 10 300 python bench.py --config cfg5 --no-cpu-baseline --headline-only > gpurun_out/b.log 2>&1
  tail -1 gpurun_out/b.log | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('four_waves=$1 blocks=$2',d['value'],d['ms_per_step'],{k.split('::')[-1]:v for k,v in d['kernel_ms'].items()})"
done; done
