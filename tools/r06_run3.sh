set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout -k 10 400 python bench.py --config dense_edge > $O/bench_dense_edge.log 2>&1 || { tail -30 $O/bench_dense_edge.log; exit 1; }
tail -1 $O/bench_dense_edge.log > $O/r06_bench_dense_edge_v2.json
echo ok
