# One gpurun call of the development loop: GPU test suite, per-layer times, a short bench line and the LDS / MFMA counters of the
# same command.  A step that is killed or times out ends the call (no further GPU step after a hang).
# usage: bash tools/gpu_check.sh <tag> [pytest-args...]   -> gpurun_out/<tag>/
tag=${1:-chk}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
step() { echo "$(date +%T) $*" | tee -a $out/progress.txt; }
ok() { [ "$1" -eq 0 ] || [ "$1" -eq 1 ]; }          # 0 = fine, 1 = test failures / bench error: keep going; anything else (124, 137, ...): stop
step pytest; timeout -k 10 900 python -m pytest tests -m gpu -q "$@" > $out/pytest.txt 2>&1; rc=$?; tail -n 30 $out/pytest.txt; ok $rc || exit $rc
step layers_f16f6; timeout -k 10 200 python tools/layer_times.py tdnn f16f6 2>&1 | grep -v amdgpu.ids > $out/layers_f16f6.txt; rc=$?; ok $rc || exit $rc
step layers_bf16x3; timeout -k 10 200 python tools/layer_times.py tdnn bf16x3 2>&1 | grep -v amdgpu.ids > $out/layers_bf16x3.txt; rc=$?; ok $rc || exit $rc
step bench; timeout -k 10 300 python bench.py --cpu-seconds 0 --no-extra > $out/bench_f16f6.json 2> $out/bench.err; rc=$?; ok $rc || exit $rc
step pmc; timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc -- python bench.py --cpu-seconds 0 --no-extra --steps 3 --warmup 1 --no-profile > $out/pmc.log 2>&1; rc=$?; ok $rc || exit $rc
python profiles/summarize_pmc.py $(find $out/pmc -name "*counter_collection.csv" | head -1) > $out/pmc_summary.txt 2>&1
rm -rf $out/pmc
step done
cat $out/layers_f16f6.txt $out/layers_bf16x3.txt; head -c 400 $out/bench_f16f6.json; grep -A8 "gemm_f16f6_kernel<7>\|w14p2_kernel<1, 3" $out/pmc_summary.txt | head -40
