set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
for dbg in none noroll novar "noroll,novar"; do
GCM_BENCH_DBG=$dbg timeout -k 10 400 python -X faulthandler bench.py --config dense_edge --no-cpu-baseline > $O/b3.log 2>&1; echo $dbg rc=$?
done
