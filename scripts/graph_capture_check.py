import sys, time, torch
sys.path.insert(0, '.')
from neural_sound_generation_amd import models as M
from neural_sound_generation_amd.train import FusedTrainStep
dev='cuda:0'
for dt, B in ((torch.bfloat16, 128), (torch.bfloat16, 16), (torch.float32, 64)):
    res = {}
    for use_graph in (False, True):
        torch.manual_seed(1)
        m = M.VQVAE(1,128,512,compute_dtype=dt).to(dev).train()
        st = FusedTrainStep(m)
        c = torch.rand(B,1,80,1024,device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        if use_graph:
            st.capture(c, warmup=2)
        else:
            st.step(c); st.step(c)
        for _ in range(3): l = st.step(c)
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(20): l = st.step(c)
        torch.cuda.synchronize(); d=(time.perf_counter()-t)/20
        res[use_graph] = (d, [x.item() for x in l], m.state_dict()['encoder.3.weight'].clone())
        print(dt, 'B', B, 'graph' if use_graph else 'eager', '%.3f ms/step' % (d*1e3), '%.2f M frames/s' % (B*1024/d/1e6), res[use_graph][1])
    print('  identical losses:', res[False][1] == res[True][1], ' identical weights:', torch.equal(res[False][2], res[True][2]))
