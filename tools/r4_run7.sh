#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
V=spiking-diffusion_amd/spkdiff/variants
cd $R
{ for pass in 1 2; do for l in lib_vt_w8 lib_vt_w12 lib_vt_w16 lib_vt_d5; do echo "== pass $pass $l"; SPKDIFF_LIB=$R/$V/$l.so python tools/convt_time.py 1024 20; SPKDIFF_LIB=$R/$V/$l.so python bench.py --workload encdec --steps 20 --warmup 5 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('encdec', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms', {k: round(v,4) for k,v in d.get('per_layer_ms', d.get('roofline',{}).get('all_kernels_avg_ms',{})).items()} )"; done; done; } > gpurun_out/r4_ab6.log 2>&1
grep -v amdgpu.ids gpurun_out/r4_ab6.log | cut -c1-400 | tail -24
SPKDIFF_LIB=$R/$V/lib_vt_w16.so python tools/vae_fp6_stress.py 10 8 2>&1 | tail -1
