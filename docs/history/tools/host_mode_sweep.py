import sys, os, time, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from act_amd import capi
L=128; PB=bench.proof_bytes(L); n=1<<int(os.environ.get("ACT_SWEEP_LOG2","18"))
h=capi.params_new("bench-org","bench-service","bench-env","2024-01-01")
eng=capi.Engine(h,L,max_batch=65536,transcript=capi.TRANSCRIPT_HOST)
sk=eng.private_key_random(bench.shake("bench-sk",64))
proofs=bench.make_inputs(eng,sk,4096)
host=np.frombuffer(proofs,np.uint8).reshape(4096,PB)
dev=torch.from_numpy(host.copy()).cuda().repeat(n//4096,1).contiguous()
st=torch.zeros(n,dtype=torch.uint8,device="cuda")
hp=torch.empty((n,PB),dtype=torch.uint8,pin_memory=True); hp.copy_(dev); hs=torch.zeros(n,dtype=torch.uint8,pin_memory=True)
torch.cuda.synchronize()
def t(fn):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter(); fn(); torch.cuda.synchronize(); return time.perf_counter()-t0
for thr in (0,):
    eng.lib.act_ctx_set_host_threads(eng.ctx, thr)
    a=t(lambda: eng.verify_spend_dev(sk,n,dev.data_ptr(),st.data_ptr())); b=t(lambda: eng.verify_spend_ptr(sk,n,capi.MEM_HOST,hp.data_ptr(),hs.data_ptr()))
    print("chunk",os.environ.get("ACT_HOST_CHUNK"),"threads",thr,"hbm",round(n/a),"hostmem",round(n/b), flush=True)
eng.set_transcript_mode(capi.TRANSCRIPT_DEVICE)
a=t(lambda: eng.verify_spend_dev(sk,n,dev.data_ptr(),st.data_ptr())); b=t(lambda: eng.verify_spend_ptr(sk,n,capi.MEM_HOST,hp.data_ptr(),hs.data_ptr()))
print("device transcripts: hbm",round(n/a),"hostmem",round(n/b), flush=True)
