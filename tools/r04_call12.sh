#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r04/bench_n1.json 2> gpurun_out/r04/bench_n1.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r04/bench_n1.err
python -c "
import json
d=json.loads(open('gpurun_out/r04/bench_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('launch_shape'), d['roofline'].get('frac_of_achievable'))
for c in d['other_configs']: print(c['workload'], round(c['ms_per_step'],5), round(c['hbm_frac'],3), c.get('launch_shape'))
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
timeout 600 python bench.py --gpus 2 --oversubscribe --no-cpu-baseline --steps 20 --warmup 3 --repeats 2 > gpurun_out/r04/bench_over2.json 2> gpurun_out/r04/bench_over2.err; echo "bench oversubscribed rc=$?"; tail -c 600 gpurun_out/r04/bench_over2.err; tail -c 1500 gpurun_out/r04/bench_over2.json
bash tools/profile_r04.sh > gpurun_out/r04/profile.log 2>&1; echo "profile rc=$?"; tail -60 gpurun_out/r04/profile.log | cut -c1-220
