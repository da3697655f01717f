import ctypes, os, sys
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"graph-conv-memory_amd")]
import torch, bench
c=bench.CONFIGS["cfg3"]; dev=torch.device("cuda",0)
lib=ctypes.CDLL(os.environ["STAMPLIB"])
B,N,F=c["B"],c["N"],c["F"]; T=128
obs=torch.rand(T,B,F,device=dev)
bits=torch.empty(T,B,4,dtype=torch.int32,device=dev)
lib.gcm_euclid_rollout_tp_decide.argtypes=[ctypes.c_void_p,ctypes.c_float,ctypes.c_void_p,ctypes.c_void_p]+[ctypes.c_int]*4+[ctypes.c_void_p]
for _ in range(3):
    rc=lib.gcm_euclid_rollout_tp_decide(obs.data_ptr(),2.0,None,bits.data_ptr(),T,B,N,F,None)
    torch.cuda.synchronize()
out=(ctypes.c_ulonglong*128)()
lib.gcm_debug_read_stamps_tp_all(out)
t0=min(out[w*8+0] for w in range(16))
print("wave ct rb | top_end chain_end copy_end store_end touch_end | barrier_out   (cycles after the first wave's round start)")
for w in range(16):
    v=[out[w*8+i]-t0 for i in range(7)]
    print(f"{w:3d} {w&3} {w>>2} | start {v[0]:6d} top {v[1]:6d} chain {v[2]:6d} copy {v[3]:6d} stage {v[4]:6d} pre-barrier {v[5]:6d} | out {v[6]:6d}")
