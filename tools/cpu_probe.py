import os, sys, time, torch
sys.path[:0]=['/root/repo/spiking-diffusion_amd','/root/repo']
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max','/sys/fs/cgroup/cpu/cpu.cfs_quota_us','/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|^CPU\\(s\\)'")
from oracle import snn_ref as ref
from spkdiff import synth
sd=synth.synth_denoiser_state(synth.MNIST)
x=torch.full((16,1,7,7),128.); t=torch.full((16,),50)
for n in (8,16,32,64,128):
    torch.set_num_threads(n)
    with torch.inference_mode():
        t0=time.perf_counter(); ref.denoiser_forward(x,t,sd,16); t1=time.perf_counter()
        ref.denoiser_forward(x,t,sd,16); t2=time.perf_counter()
    print(n,'threads: first',round(t1-t0,3),'second',round(t2-t1,3), flush=True)
    if t2-t1>20: break
