"""Where does __graft_entry__.smoke() spend its time?  (python tools/smoke_time.py on the GPU box)"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
import torch
print(f"import torch {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
print(f"cuda init {time.time() - t0:.1f} s", flush=True)
import __graft_entry__ as g
pr = cProfile.Profile()
t0 = time.time()
pr.enable(); g.smoke(); pr.disable()
print(f"smoke() {time.time() - t0:.1f} s", flush=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
print(s.getvalue()[:5000])
