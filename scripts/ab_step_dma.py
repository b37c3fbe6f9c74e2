import ctypes, sys, time, torch
sys.path.insert(0, '.')
from neural_sound_generation_amd import _lib, models as M, ops
from neural_sound_generation_amd.train import FusedTrainStep
lib = _lib.load(); lib.nsg_debug_set_gather_dma.argtypes=[ctypes.c_int]
dev='cuda:0'
for B in (64, 128):
    for on in (0, 1, 0, 1):
        lib.nsg_debug_set_gather_dma(on)
        torch.manual_seed(1)
        m = M.VQVAE(1,128,512,compute_dtype=torch.bfloat16).to(dev).train()
        st = FusedTrainStep(m)
        c = torch.rand(B,1,80,1024,device=dev)
        for _ in range(5): st.step(c)
        timer = ops.KernelTimer(); ops.KERNEL_TIMER = timer
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(20): st.step(c)
        torch.cuda.synchronize(); d=(time.perf_counter()-t)/20
        ops.KERNEL_TIMER=None
        s = timer.summary()['gather_gemm_f32']
        print('B',B,'dma',on,'%.3f ms/step'%(d*1e3), 'gather total %.3f ms/step %.0f TF'%(s['total_ms']/20, s['tflops']))
