# Full GPU check: parity tests + bench lines for every network (run on the GPU box).
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; tail -3 gpurun_out/pytest.log | cut -c1-250
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1].split('/')[-1], d['value'], 'utt/s', d['ms_per_step'], 'ms', d['tflops_algorithmic'], 'TF', '%.1e'%d['parity_rel_l2_max'], d.get('cpu_baseline') and d['cpu_baseline']['value']);print('   ', [(k['name'][:12],k['ms'],k['tflops']) for k in d['kernels']])" $1 || tail -5 gpurun_out/err.txt; }
python bench.py --network resnet_18 --batch 64 --steps 10 --cpu-seconds 6 > gpurun_out/bench_resnet.json 2>gpurun_out/err.txt; show gpurun_out/bench_resnet.json
python bench.py --network resnet_18 --batch 64 --steps 10 --cpu-seconds 0 --precision f32 > gpurun_out/bench_resnet_f32.json 2>gpurun_out/err.txt; show gpurun_out/bench_resnet_f32.json
python bench.py --network extended_tdnn --cpu-seconds 0 > gpurun_out/bench_etdnn.json 2>gpurun_out/err.txt; show gpurun_out/bench_etdnn.json
python bench.py --cpu-seconds 0 > gpurun_out/bench_tdnn.json 2>gpurun_out/err.txt; show gpurun_out/bench_tdnn.json
