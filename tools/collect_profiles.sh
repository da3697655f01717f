#!/bin/bash
# Collect the rocprofv3 evidence of a round on the MI355X box (run from the repo root through gpurun):
#   bash tools/collect_profiles.sh r06 [part ...]      parts: bench stats stats_tools pmc sq dense   (default: all)
# Kernel-trace statistics and PMC passes are separate runs (a --pmc pass never carries other trace
# domains); everything lands under gpurun_out/<tag>/ and is copied into profiles/ afterwards.
set -e -o pipefail
TAG=${1:-r05}
shift || true
PARTS=${*:-bench stats stats_tools pmc sq dense}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
stats() {   # name, script args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.log" 2>&1 < /dev/null
  cp "$(ls -t "$OUT/$name"/*/*kernel_stats.csv | head -1)" "$OUT/${TAG}_${name}_kernel_stats.csv"
  if grep -q '^{' "$OUT/$name.log"; then grep '^{' "$OUT/$name.log" | tail -1 > "$OUT/${TAG}_${name}_profiled.json"; fi   # (tools that print no JSON line leave no file)
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
      for cfg in cfg2 cfg3 cfg4 cfg5 dense_edge; do
        # (its own log name: the stats part below writes bench_$cfg.log; a failed run must not leave the previous line in place)
        if python3 bench.py --config $cfg > "$OUT/benchline_$cfg.log" 2>&1; then
          tail -1 "$OUT/benchline_$cfg.log" > "$OUT/${TAG}_bench_$cfg.json"; echo "bench $cfg done"
        else
          rm -f "$OUT/${TAG}_bench_$cfg.json"; echo "bench $cfg FAILED (see benchline_$cfg.log)"
        fi
      done ;;
    stats)   # --kernel-trace --stats of the same command (cfg4: the one-shot headline leg alone, and both legs)
      # (under rocprofv3 bench.py runs its timed region alone - the profiler's kernel statistics are the headline's)
      stats bench_cfg2 bench.py --config cfg2 --no-cpu-baseline
      stats bench_cfg3 bench.py --config cfg3 --no-cpu-baseline
      stats bench_cfg4 bench.py --config cfg4 --no-cpu-baseline
      stats bench_cfg5 bench.py --config cfg5 --no-cpu-baseline
      stats bench_dense_edge bench.py --config dense_edge --no-cpu-baseline ;;
    stats_tools)
      stats euclid_full tools/prof_euclid_full.py
      stats euclid_tp tools/prof_euclid_tp.py
      stats sparse_learned tools/prof_sparse_learned.py
      stats layered tools/prof_layered.py
      stats t256_cfg5 tools/prof_t256.py cfg5      # (the kernels of rollouts of 2 x graph_size steps: bench.py keeps no in-process trace of them)
      stats t256_cfg3 tools/prof_t256.py cfg3 ;;
    pmc)
      pmc FETCH_SIZE tools/pmc_run.py
      pmc WRITE_SIZE tools/pmc_run.py
      python3 tools/pmc_summarise.py "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$TAG" > "$OUT/${TAG}_traffic.txt"
      cp profiles/${TAG}_traffic_detail.json profiles/traffic.json "$OUT/"
      rm -rf "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE ;;
    sq)      # MFMA utilisation, LDS conflicts, issue stalls of the MFMA kernels (north_star: "MFMA utilisation (dense
             # path)"); SQ has 8 counter slots per pass; names the installed rocprofv3 does not list are dropped
      rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
      pick() { local out=""; for c in "$@"; do grep -qw "$c" "$OUT/counters_available.txt" && out="$out $c"; done; echo $out; }
      P1=$(pick SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE)
      P2=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES)
      P3=$(pick SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES)
      i=0
      for P in "$P1" "$P2" "$P3"; do
        i=$((i+1))
        [ -z "$P" ] && continue
        echo "sq pass $i: $P"
        rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/sq_$i" -- python3 tools/pmc_mfma_run.py > "$OUT/sq_$i.log" 2>&1 < /dev/null \
          || echo "sq pass $i failed (see sq_$i.log)"
      done
      python3 tools/pmc_sq_summarise.py "$TAG" "$OUT"/sq_1 "$OUT"/sq_2 "$OUT"/sq_3 > "$OUT/${TAG}_mfma_util.txt" || true
      cp profiles/${TAG}_mfma_util.json "$OUT/" || true
      cp profiles/${TAG}_mfma_util.json "$OUT/mfma_util.json" || true    # (-> profiles/mfma_util.json: what bench.py reads)
      rm -rf "$OUT"/sq_1 "$OUT"/sq_2 "$OUT"/sq_3 ;;
    dense)   # the dense-materialised regime (bench.py --config dense_edge): the general live-row kernel with every row
             # live (the column-write cached step switched off) - kernel statistics of the same bench command, then
             # HBM traffic and SQ counters of both forms from tools/pmc_dense_run.py (side profile: its own tag)
      GCM_COL_CACHE=0 python3 bench.py --config dense_edge --no-cpu-baseline > "$OUT/bench_dense_edge_general.log" 2>&1 \
        && tail -1 "$OUT/bench_dense_edge_general.log" > "$OUT/${TAG}_bench_dense_edge_general.json"
      export GCM_COL_CACHE=0
      stats bench_dense_edge_general bench.py --config dense_edge --no-cpu-baseline
      unset GCM_COL_CACHE
      pmc FETCH_SIZE tools/pmc_dense_run.py
      pmc WRITE_SIZE tools/pmc_dense_run.py
      python3 tools/pmc_summarise.py "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "${TAG}_dense" --keep-traffic-json > "$OUT/${TAG}_dense_traffic.txt"
      cp profiles/${TAG}_dense_traffic_detail.json "$OUT/"
      rm -rf "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE
      rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
      pick() { local out=""; for c in "$@"; do grep -qw "$c" "$OUT/counters_available.txt" && out="$out $c"; done; echo $out; }
      P1=$(pick SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE)
      P2=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES)
      i=0
      for P in "$P1" "$P2"; do
        i=$((i+1))
        [ -z "$P" ] && continue
        rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/dsq_$i" -- python3 tools/pmc_dense_run.py > "$OUT/dsq_$i.log" 2>&1 < /dev/null \
          || echo "dense sq pass $i failed (see dsq_$i.log)"
      done
      python3 tools/pmc_sq_summarise.py "${TAG}_dense" "$OUT"/dsq_1 "$OUT"/dsq_2 > "$OUT/${TAG}_dense_mfma_util.txt" || true
      cp profiles/${TAG}_dense_mfma_util.json "$OUT/" || true
      rm -rf "$OUT"/dsq_1 "$OUT"/dsq_2 ;;
  esac
done
echo "all done"
