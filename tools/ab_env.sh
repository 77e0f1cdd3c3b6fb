# A/B of environment settings: bash tools/ab_env.sh "VAR=val VAR2=val" "..." (bench.py, bf16x3, per-layer ms)
mkdir -p gpurun_out
for v in "$@"; do env $v timeout -k 10 300 python bench.py --cpu-seconds 0 > gpurun_out/b.json 2>gpurun_out/err.txt; python -c "
import json;d=json.load(open('gpurun_out/b.json'));print('$v', d['value'], '%.1e'%d['parity_rel_l2_max'], [(k['name'][:6],k['ms']) for k in d['kernels']])" || tail -5 gpurun_out/err.txt; done
