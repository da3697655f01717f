# A/B of two builds of libgcm_hip.so on the same box: bench dense_edge headline with each, twice
L=graph-conv-memory_amd/gcm/_lib
cp $L/libgcm_hip.so /tmp/libA.so
for r in 1 2; do
for v in A varB; do
  if [ $v = A ]; then cp /tmp/libA.so $L/libgcm_hip.so; else cp $L/libgcm_hip_$v.so $L/libgcm_hip.so; fi
  timeout -k 10 300 python bench.py --config dense_edge --no-cpu-baseline --headline-only > gpurun_out/b.log 2>&1
  tail -1 gpurun_out/b.log | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',d['value'],d['kernel_ms'])"
done; done
cp /tmp/libA.so $L/libgcm_hip.so
