# Collect the round's evidence on the GPU box: bench lines (default f16f6, bf16x3, exact f32, the other configs),
# rocprofv3 kernel trace/stats of the same command, PMC passes (separate runs, kernel-trace only).
# usage: bash tools/profile_round.sh <tag>    -> gpurun_out/<tag>/
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
step() { echo "$(date +%T) $*" >> $out/progress.txt; }
step bench; python bench.py > $out/bench_f16f6.json 2> $out/bench_f16f6.err      # the default line: TDNN family in f16f6
step bf16x3; python bench.py --precision bf16x3 --cpu-seconds 0 > $out/bench_bf16x3.json 2> $out/bench_bf16x3.err
step f32; python bench.py --precision f32 --cpu-seconds 0 --no-extra > $out/bench_f32.json 2> $out/bench_f32.err
step att; python bench.py --pooling self_attention --cpu-seconds 0 --no-extra > $out/bench_att_f16f6.json 2> $out/bench_att.err
python bench.py --pooling self_attention --precision bf16x3 --cpu-seconds 0 --no-extra > $out/bench_att_bf16x3.json 2>> $out/bench_att.err
step varlen; python bench.py --varlen --cpu-seconds 0 --no-extra > $out/bench_varlen_f16f6.json 2> $out/bench_varlen.err
python bench.py --varlen --precision bf16x3 --cpu-seconds 0 --no-extra > $out/bench_varlen_bf16x3.json 2>> $out/bench_varlen.err
step etdnn; python bench.py --network extended_tdnn --precision bf16x3 --cpu-seconds 0 --no-extra > $out/bench_etdnn_bf16x3.json 2> $out/bench_etdnn.err
python bench.py --network extended_tdnn --cpu-seconds 0 --no-extra > $out/bench_etdnn_f16f6.json 2>> $out/bench_etdnn.err
step resnet; python bench.py --network resnet_18 --batch 64 --dim 40 --cpu-seconds 4 --no-extra > $out/bench_resnet18_f16f6.json 2> $out/bench_resnet.err
python bench.py --network resnet_18 --batch 64 --dim 40 --precision bf16x3 --cpu-seconds 0 --no-extra > $out/bench_resnet18_bf16x3.json 2>> $out/bench_resnet.err
step f16; python bench.py --precision f16x3 --cpu-seconds 0 --no-extra > $out/bench_f16x3.json 2> $out/bench_f16x3.err
step att_f16; python bench.py --pooling self_attention --precision f16x3 --cpu-seconds 0 --no-extra > $out/bench_att_f16x3.json 2> $out/bench_att_f16.err
step ab; python tools/ab_options.py tail_split tdnn 2>&1 | grep -v amdgpu.ids > $out/ab_tail_split.txt 2>&1
python tools/reader_rate.py 100000 /tmp 2>&1 | grep -v amdgpu.ids > $out/reader_rate.txt
for p in f16f6 bf16x3; do python tools/layer_times.py tdnn $p 2>&1 | grep -v amdgpu.ids | tail -2; done > $out/layer_times.txt 2>&1
for n in etdnn resnet; do for p in f16f6 bf16x3; do echo "$n $p"; LT_MAX=60 python tools/layer_times.py $n $p 2>&1 | grep -v amdgpu.ids | tail -1; done; done >> $out/layer_times.txt 2>&1
step cli; python tools/cli_throughput.py 100000 2>&1 | grep -v amdgpu.ids > $out/cli_throughput.txt; python tools/cli_throughput.py 300000 2>&1 | grep -v amdgpu.ids >> $out/cli_throughput.txt; python tools/cli_throughput.py 150000 --varlen 2>&1 | grep -v amdgpu.ids >> $out/cli_throughput.txt
step host; python tools/host_profile.py 100000 2>&1 | grep -v amdgpu.ids > $out/host_profile.txt; python tools/pipeline_probe.py 2>&1 | grep -v amdgpu.ids > $out/pipeline_probe.txt
step trace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --cpu-seconds 0 --no-extra > $out/trace.log 2>&1
step pmc_sq; timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- python bench.py --cpu-seconds 0 --no-extra --steps 3 --warmup 1 --no-profile > $out/pmc_sq.log 2>&1
step pmc_fetch; timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python bench.py --cpu-seconds 0 --no-extra --steps 3 --warmup 1 --no-profile > $out/pmc_fetch.log 2>&1
step pmc_write; timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python bench.py --cpu-seconds 0 --no-extra --steps 3 --warmup 1 --no-profile > $out/pmc_write.log 2>&1
step trace_att; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_att -- python bench.py --pooling self_attention --cpu-seconds 0 --no-extra > $out/trace_att.log 2>&1
python profiles/summarize_trace.py $(find $out/trace_att -name "*kernel_trace.csv" | head -1) > $out/per_layer_summary_att.txt 2>&1
rm -rf $out/trace_att
step trace_resnet; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_resnet -- python bench.py --network resnet_18 --batch 64 --dim 40 --cpu-seconds 0 --no-extra > $out/trace_resnet.log 2>&1
python profiles/summarize_trace.py $(find $out/trace_resnet -name "*kernel_trace.csv" | head -1) > $out/per_layer_summary_resnet.txt 2>&1
rm -rf $out/trace_resnet
step summarise
python profiles/summarize_trace.py $(find $out/trace -name "*kernel_trace.csv" | head -1) > $out/per_layer_summary.txt 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
for c in sq fetch write; do python profiles/summarize_pmc.py $(find $out/pmc_$c -name "*counter_collection.csv" | head -1) > $out/pmc_${c}_summary.txt 2>&1; done
rocm-smi --showproductname > $out/device.txt 2>&1
python profiles/make_traffic.py $out f16f6 >> $out/progress.txt 2>&1
rm -rf $out/trace $out/pmc_sq $out/pmc_fetch $out/pmc_write
step done
cat $out/progress.txt; head -c 600 $out/bench_f16f6.json
