set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_rows_gpu.py -x -q -k "colcache or rows_path_vs_oracle or rows_path_matches_reference or cached_steps_vs_oracle" > $O/t_col.log 2>&1 || { tail -40 $O/t_col.log; exit 1; }
tail -3 $O/t_col.log
timeout -k 10 600 python -m pytest tests/test_full_size_gpu.py -x -q -k dense_edge > $O/t_dense.log 2>&1 || { tail -30 $O/t_dense.log; exit 1; }
tail -3 $O/t_dense.log
timeout -k 10 400 python bench.py --config dense_edge > $O/bench_dense_edge.log 2>&1 || { tail -30 $O/bench_dense_edge.log; exit 1; }
tail -1 $O/bench_dense_edge.log > $O/r06_bench_dense_edge_v2.json
echo ok
