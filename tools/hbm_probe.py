"""What this box's HBM delivers to plain streaming kernels: fill (write only), reduction (read only), copy (read + write), at
2 GB -- far beyond L2 (32 MB) and the 256 MB Infinity Cache.  The GEMMs of the encoder are write-heavy streams (N = 640: 761 MB
written, 361 MB read per launch); this is the roof they are to be read against."""
import torch

dev = "cuda"
n = 512 * 1024 * 1024          # floats = 2 GB
a = torch.empty((n,), dtype=torch.float32, device=dev)
b = torch.empty((n,), dtype=torch.float32, device=dev)


def t(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return sorted(ts)[len(ts) // 2]


gb = n * 4 / 1e12
print("fill   (write 2 GB)          : %.2f TB/s" % (gb / t(lambda: a.fill_(1.0))))
print("sum    (read 2 GB)           : %.2f TB/s" % (gb / t(lambda: a.sum())))
print("copy   (read 2 GB + write 2) : %.2f TB/s total" % (2 * gb / t(lambda: b.copy_(a))))
print("add    (read 4 GB + write 2) : %.2f TB/s total" % (3 * gb / t(lambda: torch.add(a, b, out=b))))
