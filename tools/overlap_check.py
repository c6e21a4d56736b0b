"""Given a rocprofv3 kernel_trace csv of a multi-stream run: how much kernel time overlaps another queue's kernel."""
import csv, glob, sys, collections
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"][:40]) for r in rows]
ev.sort()
ev = ev[len(ev) // 2:]                      # steady state
queues = collections.Counter(e[2] for e in ev)
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = sum(e[1] - e[0] for e in ev)
# union length
union, cur_s, cur_e = 0, None, None
for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("queues", dict(queues))
print(f"span {1e-3*(t1-t0):.0f} us, sum of kernel durations {1e-3*busy:.0f} us, union {1e-3*union:.0f} us, "
      f"overlap factor {busy/union:.2f}, idle {100*(1-union/(t1-t0)):.1f} %")
