import torch, time
x = torch.empty(int(7.4e9)//4, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(10): y.copy_(x)
torch.cuda.synchronize()
dt=(time.perf_counter()-t)/10
print("copy 7.4 GB: %.3f ms  %.2f TB/s (read+write)" % (dt*1e3, 2*x.numel()*4/dt/1e12))
t=time.perf_counter()
for _ in range(10): y.zero_()
torch.cuda.synchronize()
dt=(time.perf_counter()-t)/10
print("fill 7.4 GB: %.3f ms  %.2f TB/s (write)" % (dt*1e3, x.numel()*4/dt/1e12))
t=time.perf_counter()
for _ in range(10): s = x.sum()
torch.cuda.synchronize()
dt=(time.perf_counter()-t)/10
print("sum 7.4 GB: %.3f ms  %.2f TB/s (read)" % (dt*1e3, x.numel()*4/dt/1e12))
