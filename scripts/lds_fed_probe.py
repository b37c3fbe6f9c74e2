"""What the LDS-staged 128 x 128 bf16 tile design can reach with its memory side removed (nsg_debug_lds_fed_loop):
mode 2 = MFMA only, 1 = + fragment reads and the barrier, 0 = + the staging stores.  2560 workgroups x 18 chunks = one
3x3 conv of the benchmark."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sound_generation_amd import _lib

lib = _lib.use_diag().__enter__()      # the diagnostics library (libnsg_diag.so: switches, stamps, probe kernels) for this whole process
lib.nsg_debug_lds_fed_loop.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
blocks, chunks = 2560, 18 * 8
sink = torch.empty(blocks * 256, device="cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mode, name in ((2, "MFMA only"), (1, "+ ds_read_b128 fragments + barrier"), (0, "+ ds_write_b128 staging (the full LDS side)")):
    for _ in range(3):
        lib.nsg_debug_lds_fed_loop(blocks, chunks, mode, sink.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        lib.nsg_debug_lds_fed_loop(blocks, chunks, mode, sink.data_ptr(), st)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 5 * 1e-3
    fl = 2.0 * 128 * 128 * 64 * chunks * blocks
    print(f"{name:50s} {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TFLOP/s")

lib.nsg_debug_lds_fed_loop8.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
sink8 = torch.empty(1280 * 512, device="cuda:0")
for mode, name in ((1, "256x128 tile, 8 waves: reads + barrier"), (0, "256x128 tile, 8 waves: + staging stores")):
    for _ in range(3):
        lib.nsg_debug_lds_fed_loop8(1280, chunks, mode, sink8.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        lib.nsg_debug_lds_fed_loop8(1280, chunks, mode, sink8.data_ptr(), st)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 5 * 1e-3
    fl = 2.0 * 256 * 128 * 64 * chunks * 1280
    print(f"{name:50s} {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TFLOP/s")

lib.nsg_debug_lds_dma_loop.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
src = torch.rand(16384, device="cuda:0")
for _ in range(3):
    lib.nsg_debug_lds_dma_loop(blocks, chunks, src.data_ptr(), sink.data_ptr(), st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    lib.nsg_debug_lds_dma_loop(blocks, chunks, src.data_ptr(), sink.data_ptr(), st)
b.record()
torch.cuda.synchronize()
t = a.elapsed_time(b) / 5 * 1e-3
fl = 2.0 * 128 * 128 * 64 * chunks * blocks
print(f"{'128x128, staging by LDS-DMA (L2-resident source)':50s} {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TFLOP/s")
