"""Is tensor.to(device, non_blocking=True) from a PAGEABLE temporary safe on this ROCm stack when the stream is busy?
(Round 5: a one-in-a-few-runs NaN frame in the pipelined tracker pointed at the small host -> device copies of temporaries.)"""
import torch
dev = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=dev)
bad = 0
for trial in range(200):
    for _ in range(3):
        a = (a @ a).clamp(-1, 1)            # keeps the stream busy for a while
    src = torch.arange(64, dtype=torch.long) + trial      # pageable temporary
    d = src.to(dev, non_blocking=True)
    want = src.clone()
    del src
    junk = [torch.full((64,), -7, dtype=torch.long) for _ in range(64)]   # recycle the freed host memory right away
    torch.cuda.synchronize()
    if not torch.equal(d.cpu(), want):
        bad += 1
print("pageable temporaries corrupted in %d of 200 trials" % bad)
