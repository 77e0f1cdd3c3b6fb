# Collect the round's evidence on the GPU box: bench lines (default bf16x3 + exact f32), rocprofv3
# kernel trace/stats of the same command, PMC passes (separate runs, kernel-trace only).
set -x
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $out/bench_bf16x3.json 2> $out/bench_bf16x3.err
python bench.py --precision f32 --cpu-seconds 0 > $out/bench_f32.json 2> $out/bench_f32.err
python bench.py --pooling self_attention --cpu-seconds 0 > $out/bench_att_bf16x3.json 2> $out/bench_att.err
python bench.py --varlen --cpu-seconds 0 > $out/bench_varlen_bf16x3.json 2> $out/bench_varlen.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --cpu-seconds 0 > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --no-profile > $out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --no-profile > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --no-profile > $out/pmc_write.log 2>&1
ls -R $out | head -40
