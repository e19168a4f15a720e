import sys, time; sys.path.insert(0,'.')
import torch, bench
from baseboostdepth_amd import networks
from baseboostdepth_amd.synthetic import synthetic_batch
from oracle.step_ref import md2_step
H,W,SC=192,640,[0,1,2,3]
ms=[1]*4
torch.manual_seed(42)
models={"encoder":networks.ResnetEncoder(18,False),"pose_encoder":networks.ResnetEncoder(18,False,2)}
models["depth"]=networks.DepthDecoder(models["encoder"].num_ch_enc,SC); models["pose"]=networks.PoseDecoder(models["pose_encoder"].num_ch_enc,1,2)
opt=torch.optim.Adam([p for m in models.values() for p in m.parameters()],1e-4)
inputs=synthetic_batch(ms,H,W,SC,device="cpu",seed=42); noise=inputs.pop("noise")
for nt in (128,64,32,16,8):
    torch.set_num_threads(nt)
    md2_step(models,opt,inputs,ms,SC,H,W,noise)
    t=time.perf_counter(); md2_step(models,opt,inputs,ms,SC,H,W,noise); dt=time.perf_counter()-t
    print(nt, "threads: %.2f s/step -> %.2f img/s"%(dt,4/dt), flush=True)
