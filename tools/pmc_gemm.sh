# Counter passes focused on the GEMM's memory pipes (texture addresser, L1, VMEM/LDS issue) -- separate small
# runs (a pass with too many counters of one block aborts the profiler), kernel-trace only.
# usage: bash tools/pmc_gemm.sh <tag>   (results: gpurun_out/<tag>/pmc_*.txt)
tag=${1:-pmc}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # name counters...
  name=$1; shift
  echo "pass $name: $*" >> $out/progress.txt
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --no-profile > $out/$name.log 2>&1 || { echo "pass $name failed" >> $out/progress.txt; return 0; }
  f=$(find $out/$name -name "*counter_collection.csv" | head -1)
  python profiles/summarize_pmc.py "$f" gemm_bf16x3 > $out/pmc_$name.txt 2>&1
  rm -rf $out/$name
}
run ta1 GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
run sq1 SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
cat $out/progress.txt $out/pmc_*.txt
