import time, numpy as np, torch
dev='cuda:0'
x=np.random.rand(256,32,32,32,1).astype(np.float32)
pin=[torch.empty((128,32,32,32,1),dtype=torch.float32).pin_memory() for _ in range(2)]
ss=[torch.cuda.Stream() for _ in range(2)]
def tm(label, fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print('%-50s %.3f ms' % (label, 1e3*(time.perf_counter()-t0)/n), flush=True)
def a():
    for k in range(2): pin[k].copy_(torch.from_numpy(x[k*128:(k+1)*128]))
tm('2x host copy into pinned', a)
def b():
    for k in range(2):
        with torch.cuda.stream(ss[k]):
            pin[k].copy_(torch.from_numpy(x[k*128:(k+1)*128]))
tm('same under stream ctx', b)
def c():
    outs=[]
    for k in range(2):
        with torch.cuda.stream(ss[k]):
            pin[k].copy_(torch.from_numpy(x[k*128:(k+1)*128]))
            outs.append(pin[k].to(dev, non_blocking=True))
    for s in ss: s.synchronize()
tm('copy + async H2D, sync at end', c)
def d():
    outs=[]
    for k in range(2):
        with torch.cuda.stream(ss[k]):
            t=torch.from_numpy(x[k*128:(k+1)*128])
            outs.append(t.to(dev))
    for s in ss: s.synchronize()
tm('pageable .to() per chunk', d)
big=torch.randn(8192,8192,device=dev)
def e():
    outs=[]
    for k in range(2):
        with torch.cuda.stream(ss[k]):
            pin[k].copy_(torch.from_numpy(x[k*128:(k+1)*128]))
            outs.append(pin[k].to(dev, non_blocking=True))
            y=big@big
    for s in ss: s.synchronize()
tm('copy + async H2D + matmul each stream', e)
def f():
    for k in range(2):
        with torch.cuda.stream(ss[k]):
            y=big@big
    for s in ss: s.synchronize()
tm('matmul each stream only', f)
