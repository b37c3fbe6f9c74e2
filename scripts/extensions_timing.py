import sys, time, torch
sys.path.insert(0, '.')
from neural_sound_generation_amd import models as M
from neural_sound_generation_amd.train import FusedTrainStep
dev='cuda:0'
for dt in (torch.bfloat16, torch.float32):
    torch.manual_seed(1)
    m = M.VQVAE(1,128,512,n_speakers=7,compute_dtype=dt).to(dev).train()
    st = FusedTrainStep(m)
    B=128 if dt==torch.bfloat16 else 64
    c = torch.rand(B,1,80,1024,device=dev); g = torch.randint(0,7,(B,),device=dev)
    for _ in range(5): st.step(c,g)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): st.step(c,g)
    torch.cuda.synchronize(); dtm=(time.perf_counter()-t)/20
    print('configs[2] speaker-conditioned', dt, 'B',B, 'ms/step %.3f'%(dtm*1e3), 'frames/s %.0f'%(B*1024/dtm))
    m2 = M.VQVAE(1,128,512,ema_decay=0.99,compute_dtype=dt).to(dev).train()
    st = FusedTrainStep(m2)
    for _ in range(5): st.step(c)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): st.step(c)
    torch.cuda.synchronize(); dtm=(time.perf_counter()-t)/20
    print('EMA codebook mode', dt, 'B',B, 'ms/step %.3f'%(dtm*1e3), 'frames/s %.0f'%(B*1024/dtm))
