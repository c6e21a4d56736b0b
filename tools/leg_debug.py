"""Time one bench leg alone (debug aid): python tools/leg_debug.py prepared_tables=1 dtype=f32 graph=1"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
kw = dict(a.split("=") for a in sys.argv[1:])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
net, venv, loop = bench.build_workload(dev, 0, int(kw.get("envs", 1024)), int(kw.get("nodes", 50)), kw.get("model", "l_dgn"), "round",
                                       bool(int(kw.get("graph", 1))), 1, dtype=kw.get("dtype", "f32"),
                                       prepared_tables=bool(int(kw.get("prepared_tables", 0))))
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        loop.run(50)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(rep, "ms/step", dt / 50 * 1e3, loop.counters(), flush=True)
