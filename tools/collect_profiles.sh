#!/bin/bash
# Collect the rocprofv3 evidence of a round on the MI355X box (run from the repo root through gpurun):
#   bash tools/collect_profiles.sh r04 [part ...]      parts: bench stats stats_tools pmc   (default: all)
# Kernel-trace statistics and PMC passes are separate runs (a --pmc pass never carries other trace
# domains); everything lands under gpurun_out/<tag>/ and is copied into profiles/ afterwards.
set -e -o pipefail
TAG=${1:-r04}
shift || true
PARTS=${*:-bench stats stats_tools pmc}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
stats() {   # name, script args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.log" 2>&1 < /dev/null
  cp "$(ls -t "$OUT/$name"/*/*kernel_stats.csv | head -1)" "$OUT/${TAG}_${name}_kernel_stats.csv"
  grep '^{' "$OUT/$name.log" | tail -1 > "$OUT/${TAG}_${name}_profiled.json" || true
  rm -rf "$OUT/$name"          # (the raw traces are large; gpurun_out/ travels back)
  echo "stats $name done"
}
pmc() {     # counter, script args...
  local ctr=$1; shift
  rocprofv3 --pmc "$ctr" --kernel-trace --output-format csv -d "$OUT/pmc_$ctr" -- python3 "$@" > "$OUT/pmc_$ctr.log" 2>&1 < /dev/null
  echo "pmc $ctr done"
}
for part in $PARTS; do
  case $part in
    bench)   # the JSON line of every config, unprofiled
      for cfg in cfg2 cfg3 cfg4 cfg5; do
        python3 bench.py --config $cfg > "$OUT/bench_$cfg.log" 2>&1 && tail -1 "$OUT/bench_$cfg.log" > "$OUT/${TAG}_bench_$cfg.json"
        echo "bench $cfg done"
      done ;;
    stats)   # --kernel-trace --stats of the same command (cfg4: the one-shot headline leg alone, and both legs)
      # (under rocprofv3 bench.py runs its timed region alone - the profiler's kernel statistics are the headline's)
      stats bench_cfg2 bench.py --config cfg2 --no-cpu-baseline
      stats bench_cfg3 bench.py --config cfg3 --no-cpu-baseline
      stats bench_cfg4 bench.py --config cfg4 --no-cpu-baseline
      stats bench_cfg5 bench.py --config cfg5 --no-cpu-baseline ;;
    stats_tools)
      stats euclid_full tools/prof_euclid_full.py
      stats sparse_learned tools/prof_sparse_learned.py
      stats layered tools/prof_layered.py ;;
    pmc)
      pmc FETCH_SIZE tools/pmc_run.py
      pmc WRITE_SIZE tools/pmc_run.py
      python3 tools/pmc_summarise.py "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$TAG" > "$OUT/${TAG}_traffic.txt"
      cp profiles/${TAG}_traffic_detail.json profiles/traffic.json "$OUT/"
      rm -rf "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE ;;
  esac
done
echo "all done"
