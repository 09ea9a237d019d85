import sys; sys.path.insert(0,'.')
import torch, numpy as np, theora_amd
from theora_amd import _lib
L=_lib.load()
n=2_000_000
x=torch.randint(-255,256,(n,64),dtype=torch.int16,device='cuda')
dq=torch.from_numpy(np.clip(np.arange(64)*3+16,8,4096).astype(np.uint16)).cuda()
s=torch.cuda.current_stream()
def timed(fn,reps=10):
    fn(); torch.cuda.synchronize()
    L.thip_set_batch_stream(s.cuda_stream,0)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); torch.cuda.synchronize(); L.thip_set_batch_stream(None,1)
    return e0.elapsed_time(e1)*1e-3/reps
y=torch.empty_like(x); q=torch.empty_like(x); nz=torch.empty(n,dtype=torch.int32,device='cuda')
t=timed(lambda: L.thip_enc_fdct8x8_batch(y.data_ptr(),x.data_ptr(),n))
print("fdct  %.1f us  %.0f GB/s  %.1f Gblk/s"%(t*1e6, n*256/t/1e9, n/t/1e9))
t=timed(lambda: L.thip_enc_quantize_batch(q.data_ptr(),nz.data_ptr(),y.data_ptr(),dq.data_ptr(),n))
print("quant %.1f us  %.0f GB/s  %.1f Gblk/s"%(t*1e6, n*260/t/1e9, n/t/1e9))
lz=torch.full((n,),63,dtype=torch.int32,device='cuda')
t=timed(lambda: L.thip_idct8x8_batch(q.data_ptr(),y.data_ptr(),lz.data_ptr(),n))
print("idct  %.1f us  %.0f GB/s  %.1f Gblk/s"%(t*1e6, n*256/t/1e9, n/t/1e9))
