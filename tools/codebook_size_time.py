"""The dense 100-step reverse process at B = 256 for several --codebook_size values (R/main.py:58): ms per sample() call (one hipGraph replay)."""
import sys, time
sys.path[:0]=["spiking-diffusion_amd","."]
sys.argv=["bench.py"]
import torch, bench
dev=torch.device("cuda",0)
for name in ("mnist","mnist_k256","mnist_k512","mnist_k100"):
    cfg=bench.path_config(name)
    model,den,ab=bench.build_models(dev,16,cfg)
    ab.n_samples=256; ab.skip_untouched=False
    torch.manual_seed(1)
    for _ in range(2): ab.sample(temp=1.0,sample_steps=100)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(3): ab.sample(temp=1.0,sample_steps=100)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/3
    print(name, "dense ms per 100-step sample B=256:", round(dt*1e3,2), flush=True)
    ab._graphs.clear(); del model,den,ab
