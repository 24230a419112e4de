import time, numpy as np, torch
dev='cuda:0'
x=np.random.rand(256,32,32,32,1).astype(np.float32)
print('threads', torch.get_num_threads())
def t(fn,n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3*(time.perf_counter()-t0)/n
d=torch.empty(x.shape,dtype=torch.float32,device=dev)
pin=torch.empty(x.shape,dtype=torch.float32).pin_memory()
pin2=torch.empty(x.shape,dtype=torch.float32).pin_memory()
xt=torch.from_numpy(x)
print('pageable H2D .to()        %.3f ms'%t(lambda: torch.from_numpy(x).to(dev)))
print('pageable H2D copy_        %.3f ms'%t(lambda: d.copy_(xt)))
print('host memcpy -> pinned     %.3f ms'%t(lambda: pin.copy_(xt)))
print('pinned H2D                %.3f ms'%t(lambda: d.copy_(pin,non_blocking=True)))
def staged():
    pin.copy_(xt); d.copy_(pin,non_blocking=True)
print('staged H2D                %.3f ms'%t(staged))
def staged4():
    for i in range(4):
        pin[i*64:(i+1)*64].copy_(xt[i*64:(i+1)*64]); d[i*64:(i+1)*64].copy_(pin[i*64:(i+1)*64],non_blocking=True)
print('staged H2D 4 chunks       %.3f ms'%t(staged4))
print('pageable D2H .cpu().numpy %.3f ms'%t(lambda: d.cpu().numpy()))
def d2h_staged():
    pin2.copy_(d,non_blocking=True); torch.cuda.synchronize(); out=np.empty(x.shape,np.float32); torch.from_numpy(out).copy_(pin2); return out
print('staged D2H + memcpy       %.3f ms'%t(d2h_staged))
def d2h_pin_only():
    pin2.copy_(d,non_blocking=True); torch.cuda.synchronize()
print('pinned D2H only           %.3f ms'%t(d2h_pin_only))
print('np.empty + memcpy out     %.3f ms'%t(lambda: torch.from_numpy(np.empty(x.shape,np.float32)).copy_(pin2)))
s1,s2=torch.cuda.Stream(),torch.cuda.Stream()
def duplex():
    with torch.cuda.stream(s1): d.copy_(pin,non_blocking=True)
    with torch.cuda.stream(s2): pin2.copy_(d,non_blocking=True)
print('H2D || D2H (2 streams)    %.3f ms'%t(duplex))
u8=torch.empty(x.shape,dtype=torch.uint8,device=dev); pin8=torch.empty(x.shape,dtype=torch.uint8).pin_memory()
def d2h_u8():
    pin8.copy_(u8,non_blocking=True); torch.cuda.synchronize()
print('pinned D2H uint8          %.3f ms'%t(d2h_u8))
