import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import bench, __graft_entry__ as e
pkg = e.load_package()
t0 = time.perf_counter()
ctx = bench.make_context(pkg, 512, 8192, 0, 65536)
t1 = time.perf_counter()
x = torch.from_numpy(bench.synth_block(512, 8192, 192000)).cuda(); y = torch.empty_like(x)
t2 = time.perf_counter()
ctx.process_device(x.data_ptr(), y.data_ptr(), 8192, 192000); ctx.synchronize()
t3 = time.perf_counter()
ctx.process_device(x.data_ptr(), y.data_ptr(), 8192, 192000); ctx.synchronize()
t4 = time.perf_counter()
print("make_context %.2f s (python builds 1024 IRs + appends 4096 units), input %.2f s, first step (plan + IR spectra + state) %.3f s, second step %.4f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
t5 = time.perf_counter(); ctx.close(); print("close %.3f s" % (time.perf_counter() - t5))
