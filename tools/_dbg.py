import sys; sys.path.insert(0, "/root/repo")
import torch
from asr_chinese_e2e_amd import kernels as K
torch.manual_seed(0)
B,H,Tq,dk=32,8,500,64
d=H*dk
for Tk in (500, 64):
    q=torch.randn(B*Tq,d,device="cuda").bfloat16()
    kv=torch.randn(B*Tk,2*d,device="cuda").bfloat16()
    klen=torch.full((B,),Tk,dtype=torch.int32,device="cuda")
    for _ in range(5):
        o,lse=K.sdpa_fwd(q,kv[:,:d],kv[:,d:],klen,B,H,Tq,Tk,dk,False,-1)
    torch.cuda.synchronize()
    print(Tk,[int(x) for x in lse.flatten()[:34].tolist()])
