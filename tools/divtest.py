import sys; sys.path.insert(0,'.')
import torch
from baseboostdepth_amd import ops
from baseboostdepth_amd._lib import ptr
be=ops.default_backend()
bad=torch.zeros(10,dtype=torch.int32,device='cuda')
be.run("bbd_selftest_div", bad, 2048, 256, 0x80000001, ptr(bad))
torch.cuda.synchronize(); print("normal[q0,q1,div,const,div9] extreme[...]:", bad.tolist())
