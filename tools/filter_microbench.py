import os, sys, time, json
import numpy as np
ROOT="/root/repo"
sys.path[:0]=[ROOT, ROOT+"/gp-ann_amd", ROOT+"/oracle"]
import torch, pygpann
dev=torch.device("cuda",0)
n,d,nq,k=int(os.environ.get("N","20000000")),128,250,10
st=torch.cuda.current_stream().cuda_stream
pts=torch.empty((n,d),dtype=torch.float32,device=dev)
pygpann.synth_rows_device(pts.data_ptr(),555,7,1024,0.35,0,n,d,st)
q=torch.empty((nq,d),dtype=torch.float32,device=dev)
pygpann.synth_rows_device(q.data_ptr(),556,7,1024,0.35,0,nq,d,st)
torch.cuda.synchronize()
sh=pygpann.Shards(pts.data_ptr(),[np.arange(n,dtype=np.uint32)],0,0,n=n,d=d)
sh.set_scan_mode(1)
oi=torch.empty((nq,1,k),dtype=torch.int32,device=dev); od=torch.empty((nq,1,k),dtype=torch.float32,device=dev)
probes=np.zeros((nq,1),np.int32)
for dbg in os.environ.get("DBGS","0,1,2,3,4,7").split(","):
    os.environ["GPANN_FILTER_DEBUG"]=dbg
    for rep in range(3):
        ms=sh.scan_buckets_device(q.data_ptr(),nq,k,probes,1,oi.data_ptr(),od.data_ptr(),st,timed=True)
    s=sh.stats()
    print("debug",dbg,"scan_ms",round(s["scan_ms"],3),"filter_pairs",s["filter_pairs"], "units", s["units"], "cand", s["filter_candidates"], flush=True)
