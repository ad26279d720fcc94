#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
V=spiking-diffusion_amd/spkdiff/variants
cd $R
python -m pytest tests -m gpu -q -x -k "vae or f2_ or f3_ or f4_ or encode or decode or config3 or full_size_properties or main_py or collapsed" 2>&1 | tail -6 > gpurun_out/r4_gputest6.log
grep -v PARITY gpurun_out/r4_gputest6.log | tail -4 | cut -c1-300
python tools/vae_fp6_stress.py 60 8 2>&1 | tail -3
{ for pass in 1 2; do for l in lib_vt_d4_t3 lib_vt_d4_t2 lib_vt_d5; do echo "== pass $pass $l"; SPKDIFF_LIB=$R/$V/$l.so python tools/convt_time.py 1024 20; SPKDIFF_LIB=$R/$V/$l.so python bench.py --workload encdec --steps 20 --warmup 5 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('encdec', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms', {k: round(v,4) for k,v in d.get('per_layer_ms', d.get('roofline',{}).get('all_kernels_avg_ms',{})).items()} )"; done; done; } > gpurun_out/r4_ab5.log 2>&1
grep -v amdgpu.ids gpurun_out/r4_ab5.log | cut -c1-400 | tail -14
