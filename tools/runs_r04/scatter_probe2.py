"""round 4 probe 2: do narrow scattered writes (an array that fits the 256 MB Infinity Cache) cost less than 8-byte ones?"""
import time
import torch

dev = torch.device("cuda:0")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


n = 100_000_000
perm = torch.randperm(n, device=dev)
for dt in (torch.int64, torch.int32, torch.int16, torch.int8):
    vals = torch.ones(n, dtype=dt, device=dev)
    out = torch.empty_like(vals)
    print("scatter of %-12s (%4d MB target): %.2f ms   sequential copy: %.2f ms" % (dt, out.numel() * out.element_size() // 2**20,
                                                                                  timed(lambda: out.index_copy_(0, perm, vals)), timed(lambda: out.copy_(vals))))
# 8-byte counts, but the random places confined to windows of W elements (W x 8 bytes of the output live at a time)
vals = torch.ones(n, dtype=torch.int64, device=dev)
out = torch.empty_like(vals)
for w in (2**16, 2**20, 2**22, 2**24, 25_000_000, 50_000_000):
    base = (torch.arange(n, device=dev) // w) * w
    local = (torch.rand(n, device=dev) * w).long().clamp_(max=w - 1)
    wperm = torch.clamp(base + local, max=n - 1)
    print("int64 scatter, random within windows of %9d elements (%5d MB): %.2f ms" % (w, w * 8 // 2**20, timed(lambda: out.index_copy_(0, wperm, vals))))
    del base, local, wperm
