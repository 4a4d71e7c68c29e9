import sys, ctypes as C
sys.path.insert(0,'/root/repo')
import torch, numpy as np
import libbtbb_amd as bt
bt.init(2); lib=bt.lib()
nw=1<<29
hs=C.c_void_p(torch.cuda.current_stream().cuda_stream)
d=torch.empty(nw,dtype=torch.int64,device='cuda')
bt.check(lib.btbbx_synth_device(d.data_ptr(),0,nw,5,4096,0x9E8B33,4,hs))
cap=1<<24
h=torch.empty(cap*2,dtype=torch.int64,device='cuda'); c=torch.zeros(1,dtype=torch.int32,device='cuda')
for me in (2, 0, 4):
    def run():
        c.zero_(); bt.check(lib.btbbx_scan_device(d.data_ptr(),nw,nw,1,nw*64-63,0x9E8B33,me,h.data_ptr(),cap,c.data_ptr(),hs))
    run(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): run()
    b.record(); torch.cuda.synchronize()
    print("known-LAP 4 GiB max_err", me, "ms", round(a.elapsed_time(b)/5,3), "hits", int(c.item()))
