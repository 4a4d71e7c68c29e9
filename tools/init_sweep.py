import sys, ctypes as C, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, numpy as np
import libbtbb_amd as bt
lib=bt.lib()
nw=1<<27
out={}
import os
for n in [int(x) for x in os.environ.get('SWEEP_N','0,1,2,3,4,5').split(',')]:
    lib.btbbx_shutdown(); bt.init(n)
    hs=C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d=torch.empty(nw,dtype=torch.int64,device='cuda')
    bt.check(lib.btbbx_synth_device(d.data_ptr(),0,nw,5,4096,-1,7,hs))
    cap=1<<22
    h=torch.empty(cap*2,dtype=torch.int64,device='cuda'); c=torch.zeros(1,dtype=torch.int32,device='cuda')
    def run():
        c.zero_()
        bt.check(lib.btbbx_scan_device(d.data_ptr(),nw,nw,1,nw*64-63,bt.LAP_ANY,n,h.data_ptr(),cap,c.data_ptr(),hs))
    run(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record(); run(); run(); b.record(); torch.cuda.synchronize()
    ms=a.elapsed_time(b)/2
    out[n]={"ms_per_GiB":round(ms,3),"Gbit_s":round(nw*64/ms/1e6,1),"hits":int(c.item())}
print(json.dumps(out))
