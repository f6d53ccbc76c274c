import torch, time
x = torch.empty(1 << 30, dtype=torch.bfloat16, device="cuda").normal_()
for fn, name, nbytes in ((lambda: x.sum(), "sum(read 2 GiB)", 2 << 30), (lambda: x.clone(), "clone(read+write 4 GiB)", 4 << 30)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, "%.2f TB/s" % (nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12))
