"""LDS-DMA weight-stream rate of one workgroup per CU against the bytes kept in flight (tools/exp/dma_ring.hip)."""
import ctypes, os, torch
so = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdma_ring.so"))
so.run_ring.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
frags = 2048                                   # 2 MB image
img = torch.randint(0, 255, (frags * 1024,), dtype=torch.uint8, device="cuda")
out = torch.zeros((4,), device="cuda")
blocks, reps = 256, 9                          # 9 passes over the image per workgroup ~ the FFN kernel's 9.08 rounds
for mode, name in ((0, "2 x 64 KB stages, 1 in flight"), (1, "4 x 32 KB stages, 3 in flight"), (2, "8 x 16 KB stages, 7 in flight")):
    for _ in range(2):
        so.run_ring(mode, img.data_ptr(), frags, out.data_ptr(), reps, blocks, None)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); so.run_ring(mode, img.data_ptr(), frags, out.data_ptr(), reps, blocks, None); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[2]
    print("%-32s %8.1f us for %.2f GB -> %.2f TB/s (%.1f GB/s per CU)" % (name, t, blocks * reps * frags * 1024 / 1e9, blocks * reps * frags * 1024 / t / 1e6, reps * frags * 1024 / t / 1e3))
