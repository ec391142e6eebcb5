"""round 4 probe: what do random 8-byte scatters / gathers and a radix sort cost on this GPU? (torch kernels as the yardstick)"""
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


for n in (100_000_000, 300_000_000):
    perm = torch.randperm(n, device=dev)
    vals = torch.arange(n, dtype=torch.int64, device=dev)
    out = torch.empty_like(vals)
    print("n=%d" % n)
    print("  scatter out[perm]=vals (random 8B writes): %.2f ms" % timed(lambda: out.index_copy_(0, perm, vals)))
    print("  gather  out=vals[perm] (random 8B reads):  %.2f ms" % timed(lambda: torch.index_select(vals, 0, perm, out=out)))
    perm32 = perm.to(torch.int32)
    print("  copy 8B sequential: %.2f ms" % timed(lambda: out.copy_(vals)))
    keys = torch.randint(0, 1 << 24, (n,), dtype=torch.int32, device=dev)
    print("  torch.sort int32 keys (24 random bits) with indices: %.2f ms" % timed(lambda: torch.sort(keys), reps=3))
    keys64 = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device=dev)
    print("  torch.sort int64 keys with indices: %.2f ms" % timed(lambda: torch.sort(keys64), reps=3))
    # windowed scatter: positions random only within windows of 64 Ki elements (512 KiB)
    win = 65536
    base = (torch.arange(n, device=dev) // win) * win
    local = torch.randint(0, win, (n,), device=dev)
    wperm = torch.clamp(base + local, max=n - 1)
    print("  windowed scatter (random within 512 KiB windows): %.2f ms" % timed(lambda: out.index_copy_(0, wperm, vals)))
    print("  windowed gather: %.2f ms" % timed(lambda: torch.index_select(vals, 0, wperm, out=out)))
    rows = torch.empty((n, 31), dtype=torch.uint8, device=dev)
    rout = torch.empty_like(rows) if n <= 100_000_000 else None
    if rout is not None:
        print("  gather 31-byte rows by random perm: %.2f ms" % timed(lambda: torch.index_select(rows, 0, perm, out=rout), reps=3))
    del perm, vals, out, keys, keys64, base, local, wperm, rows, rout
    torch.cuda.empty_cache()
