"""Does a working set inside the 256 MiB Infinity Cache stream faster than one in HBM?  copy / read / write rates of
torch kernels against the buffer size (each op repeated on the SAME buffers, so a small set stays cache-resident)."""
import torch, time
for mb in (16, 32, 64, 96, 128, 192, 256, 512, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.ones(n, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
    reps = max(10, int(20000 / mb))
    out = []
    for name, fn, nbytes in (("copy", lambda: y.copy_(x), 2 * n * 4), ("fill", lambda: y.zero_(), n * 4), ("sum", lambda: x.sum(), n * 4)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / reps
        out.append("%s %.2f TB/s (%.1f us)" % (name, nbytes / dt / 1e12, dt * 1e6))
    print("%5d MiB per buffer: %s" % (mb, "  ".join(out)), flush=True)
