"""Elimination forms with the active-list step tail (one launch per slot: conv6 on counts + token update) against the two launches it replaces: ms per sample() call, tokens equal."""
import sys, time
sys.path[:0]=["spiking-diffusion_amd","."]
sys.argv=["bench.py"]
import torch, bench
dev=torch.device("cuda",0)
for B, steps in ((256,100),(16,49),(64,100)):
    model,den,ab=bench.build_models(dev,16)
    ab.n_samples=B
    res={}
    toks={}
    for name,(sk,li,ta) in {"elim+lists tail":(True,True,True),"elim+lists two":(True,True,False),"elim tail":(True,False,True),"elim two":(True,False,False)}.items():
        ab.skip_untouched, ab.list_positions, ab.step_tail_in_elimination = sk, li, ta
        torch.manual_seed(1)
        for _ in range(2): ab.sample(temp=1.0,sample_steps=steps)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(5): tok=ab.sample(temp=1.0,sample_steps=steps)
        torch.cuda.synchronize(); res[name]=(time.perf_counter()-t0)/5*1e3
        torch.manual_seed(7); toks[name]=ab.sample(temp=1.0,sample_steps=steps).cpu()
    same=all(torch.equal(v, toks["elim two"]) for v in toks.values())
    print(f"B={B} steps={steps}:", " | ".join(f"{k} {v:.2f} ms" for k,v in res.items()), "tokens equal", same, flush=True)
    ab._graphs.clear(); del model,den,ab
