#!/bin/bash
# Collect the rocprofv3 evidence of a round on the MI355X box (run from the repo root through gpurun):
#   bash tools/collect_profiles.sh r02
# Kernel-trace statistics and PMC passes are separate runs (a --pmc pass never carries other trace
# domains); everything lands under gpurun_out/<tag>/ and is copied into profiles/ afterwards.
set -e -o pipefail
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
stats() {   # name, script args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.log" 2>&1 < /dev/null
  cp "$(ls -t "$OUT/$name"/*/*kernel_stats.csv | head -1)" "$OUT/${TAG}_${name}_kernel_stats.csv"
  rm -rf "$OUT/$name"          # (the raw traces are large; gpurun_out/ travels back)
  echo "stats $name done"
}
pmc() {     # counter, name, script args...
  local ctr=$1 name=$2; shift 2
  rocprofv3 --pmc "$ctr" --kernel-trace --output-format csv -d "$OUT/${name}_$ctr" -- python3 "$@" > "$OUT/${name}_$ctr.log" 2>&1 < /dev/null
  echo "pmc $ctr $name done"
}
python3 bench.py > "$OUT/bench.log" 2>&1 && tail -1 "$OUT/bench.log" > "$OUT/${TAG}_bench.json"
echo "bench done"
stats bench bench.py --no-cpu-baseline
grep '^{' "$OUT/bench.log" | tail -1 > "$OUT/${TAG}_bench_profiled.json"
stats cfg4 tools/bench_configs.py cfg4
stats cfg3 tools/bench_configs.py "cfg3 EuclideanEdge(2.0) cross-batch"
stats cfg5 tools/bench_configs.py cfg5
pmc FETCH_SIZE dense tools/pmc_run.py
pmc WRITE_SIZE dense tools/pmc_run.py
pmc FETCH_SIZE cfg4 tools/bench_configs.py cfg4-oneshot
pmc WRITE_SIZE cfg4 tools/bench_configs.py cfg4-oneshot
python3 tools/pmc_summarise.py "$OUT/dense_FETCH_SIZE" "$OUT/dense_WRITE_SIZE" "$TAG" > "$OUT/${TAG}_traffic.txt"
python3 tools/pmc_summarise.py "$OUT/cfg4_FETCH_SIZE" "$OUT/cfg4_WRITE_SIZE" "${TAG}_cfg4_oneshot" --keep-traffic-json > "$OUT/${TAG}_cfg4_traffic.txt"
cp profiles/${TAG}_traffic_detail.json profiles/${TAG}_cfg4_oneshot_traffic_detail.json profiles/traffic.json "$OUT/"
rm -rf "$OUT"/dense_FETCH_SIZE "$OUT"/dense_WRITE_SIZE "$OUT"/cfg4_FETCH_SIZE "$OUT"/cfg4_WRITE_SIZE
python3 tools/bench_configs.py > "$OUT/${TAG}_configs.jsonl.log" 2>&1 && grep '^{' "$OUT/${TAG}_configs.jsonl.log" > "$OUT/${TAG}_configs.jsonl"
echo "all done"
