import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neural_sound_generation_amd import ops
dev="cuda:0"; bf=torch.bfloat16
B,H,W,C=1,4,8,16
torch.manual_seed(0)
x=torch.randn(B,H,W,C,device=dev).to(bf); dy=torch.randn(B,H,W,C,device=dev).to(bf)
for k,p in ((1,0),(3,1)):
    d32=ops.conv_desc(B,H,W,C,C,k,1,p); d16=ops.conv_desc(B,H,W,C,C,k,1,p,dtype=bf)
    w32,_=ops.conv_wgrad(d32,x.float(),dy.float(),(C,C,k,k),want_bias=False)
    w16,_=ops.conv_wgrad(d16,x,dy,(C,C,k,k),want_bias=False)
    print("k",k,"max diff",(w32-w16).abs().max().item(),"scale",w32.abs().max().item())
    if k==1:
        ref=(dy.float().view(-1,C).t()@x.float().view(-1,C))
        print(" fp32 vs torch",(w32.view(C,C)-ref).abs().max().item()," bf16 vs torch",(w16.view(C,C)-ref).abs().max().item())
        print(" ratio sample", (w16.view(C,C)/ref)[:2,:6])
        # is it a permutation? compare sorted values
        print(" sorted equal?", torch.allclose(w16.flatten().sort()[0], ref.flatten().sort()[0], atol=1e-3))
        ref2=(dy.float().view(-1,C).t()@x.float().view(-1,C))
        # try x columns permuted pairs
        xp=x.float().view(-1,C); 
        for name,perm in (("swap pairs",[1,0,3,2,5,4,7,6,9,8,11,10,13,12,15,14]),):
            r=(dy.float().view(-1,C).t()@xp[:,perm]); print(name,(w16.view(C,C)-r).abs().max().item())
