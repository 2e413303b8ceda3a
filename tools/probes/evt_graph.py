import torch, time
dev = torch.device('cuda', 0)
x = torch.randn(4096, 4096, device=dev)
s = torch.cuda.Stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    for _ in range(3):
        y = x @ x
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        e0.record()
        y = x @ x
        y = y @ x
        e1.record()
    for k in range(3):
        g.replay()
        torch.cuda.synchronize()
        print('replay', k, 'elapsed ms', e0.elapsed_time(e1))
except Exception as ex:
    print('FAILED', type(ex).__name__, ex)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); y = x @ x; y = y @ x; t1.record(); torch.cuda.synchronize(); print('eager ms', t0.elapsed_time(t1))
