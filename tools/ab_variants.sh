# A/B timing of GEMM variants in one process environment each (bench.py, bf16x3).  usage: bash tools/ab_variants.sh "TILE SCHED DIAG" ...
mkdir -p gpurun_out
for v in "$@"; do set -- $v; XVEC_GEMM_TILE=$1 XVEC_GEMM_DIAG=$3 timeout -k 10 300 python bench.py --precision ${4:-bf16x3} --cpu-seconds 0 > gpurun_out/b.json 2>gpurun_out/err.txt; python -c "
import json;d=json.load(open('gpurun_out/b.json'));print('tile $1 sched $2 diag $3 ${4:-bf16x3}', d['value'], '%.1e'%d['parity_rel_l2_max'], [(k['name'][:6],k['ms']) for k in d['kernels']])" || tail -5 gpurun_out/err.txt; done
