"""dev tool: measured device copy bandwidth for one octave-0 plane (the practical 'read a plane, write a
plane' floor that k_blur competes with).  torch is only used as a convenient memcpy + event timer."""
import torch
W, H = 3840, 2160
for nbuf in (2, 8, 24):        # 2: ping-pong inside the 256 MB Infinity Cache; 24: ~800 MB working set
    bufs = [torch.rand(H, W, device="cuda") for _ in range(nbuf)]
    for _ in range(3):
        for i in range(nbuf - 1): bufs[i + 1].copy_(bufs[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(1, 200 // (nbuf - 1))
    e0.record()
    for _ in range(reps):
        for i in range(nbuf - 1): bufs[i + 1].copy_(bufs[i])
    e1.record(); torch.cuda.synchronize()
    n = reps * (nbuf - 1)
    us = e0.elapsed_time(e1) * 1e3 / n
    print("buffers %2d: copy %.2f us per plane  -> %.0f GB/s (read+write)" % (nbuf, us, 2 * W * H * 4 / us / 1e3))
