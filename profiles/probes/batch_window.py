import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, __graft_entry__ as entry
pkg = entry.load_package()
nch, sr, blocks = 512, 192000, 128
ctx = bench.make_context(pkg, nch, 8192, 0, 65536)
ctx.set_window(int(os.environ.get("W", "1")))
files = bench.batch_files(nch, sr, blocks)
call, outs = ctx.batch_prepared(files, sr, "lpcm24")
call(); call()
t0 = time.perf_counter(); call(); print("run: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
ctx.close()
