import sys, ctypes; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import numpy as np, __graft_entry__ as e
pkg=e.load_package()
nch=256
ctx=pkg.Context(nch,8192)
for c in range(nch): ctx.append_unit(c,"chorus")
a,b=ctx.alloc(nch,8192),ctx.alloc(nch,8192)
a.upload(np.random.default_rng(0).uniform(-0.5,0.5,(nch,8192)))
for _ in range(5): ctx.process_device(a,b,8192,192000)
ctx.synchronize()
buf=(ctypes.c_ulonglong*(64*64))()
pkg.lib().gdg_debug_stamps(buf)
st=np.array(buf[:],dtype=np.int64)[:1024].reshape(4,16,16)
for blk in range(2):
    print("block",blk)
    for w in (0,1,7,15):
        r=st[blk,w]; print(" wave",w," ".join("%6d"%(r[i]-r[0]) for i in (1,2,3,4,5,6,13,7,8,11,12)))
