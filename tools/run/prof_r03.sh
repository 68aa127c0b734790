#!/bin/bash
# GPU box: round-3 evidence.  (1) kernel trace + stats of the bench step, (2) HBM-side byte counters (two passes:
# FETCH_SIZE and WRITE_SIZE do not fit one), (3) matrix-pipe / LDS counters.  --pmc only ever with --kernel-trace.
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs"
rm -rf $R/gpurun_out/prof_r03 $R/gpurun_out/pmc_r03_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03 -- $BENCH > $R/gpurun_out/prof_r03_bench.log 2>&1 || { tail -5 $R/gpurun_out/prof_r03_bench.log | cut -c1-300; exit 1; }
tail -1 $R/gpurun_out/prof_r03_bench.log | cut -c1-200
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
pick() { for c in "$@"; do grep -q -w "$c" $R/gpurun_out/counters.txt && printf "%s " "$c"; done; }
SETS=("FETCH_SIZE" "WRITE_SIZE" "$(pick SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE)" "$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_COEXEC_CYCLES)")
i=0
for s in "${SETS[@]}"; do
  [ -z "$s" ] && continue
  echo "pmc set $i: $s"
  timeout -k 10 300 rocprofv3 --pmc $s --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r03_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/pmc_r03_$i.log 2>&1 || { echo "set $i failed"; tail -3 $R/gpurun_out/pmc_r03_$i.log | cut -c1-300; }
  i=$((i+1))
done
find $R/gpurun_out/prof_r03 $R/gpurun_out/pmc_r03_* -name "*.csv" | head -20
