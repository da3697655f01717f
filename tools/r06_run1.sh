set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_full_size_gpu.py -x -q -k dense_edge > $O/t_dense.log 2>&1 || { tail -30 $O/t_dense.log; exit 1; }
tail -3 $O/t_dense.log
timeout -k 10 400 python bench.py --config dense_edge > $O/bench_dense_edge.log 2>&1 || { tail -30 $O/bench_dense_edge.log; exit 1; }
tail -1 $O/bench_dense_edge.log > $O/r06_bench_dense_edge.json
timeout -k 10 300 python bench.py --config cfg2 > $O/bench_cfg2.log 2>&1 || { tail -30 $O/bench_cfg2.log; exit 1; }
tail -1 $O/bench_cfg2.log > $O/r06_bench_cfg2.json
echo ok
