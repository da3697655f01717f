set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_rows_gpu.py tests/test_full_size_gpu.py -x -q -k "colcache or dense or rows_path_vs_oracle or rows_path_matches_reference" > $O/t_col.log 2>&1 || { tail -40 $O/t_col.log; exit 1; }
tail -2 $O/t_col.log
timeout -k 10 400 python bench.py --config dense_edge --no-cpu-baseline > $O/bench_dense_edge.log 2>&1 || { tail -30 $O/bench_dense_edge.log; exit 1; }
tail -1 $O/bench_dense_edge.log > $O/r06_bench_dense_edge_v4.json
python -c "
import json;d=json.load(open('$O/r06_bench_dense_edge_v4.json'));print(d['value'],d['ms_per_step'],d['kernel_ms'],d['variants'])"
