import torch, time
x = torch.zeros(64, device="cuda")
def run(n):
    for _ in range(n): x.add_(1.0)
run(100); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t=time.perf_counter(); e0.record(); run(5000); e1.record(); torch.cuda.synchronize(); w=time.perf_counter()-t
print("eager: gpu %.2f us/op, wall %.2f us/op" % (e0.elapsed_time(e1)/5000*1e3, w/5000*1e6))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run(10)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        run(1000)
torch.cuda.synchronize()
for _ in range(2): g.replay()
torch.cuda.synchronize()
t=time.perf_counter(); e0.record(); 
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize(); w=time.perf_counter()-t
print("graph: gpu %.2f us/op, wall %.2f us/op" % (e0.elapsed_time(e1)/5000*1e3, w/5000*1e6))
