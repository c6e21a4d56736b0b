"""Launch-by-launch account of one graph-replayed step: python tools/gap_trace.py <kernel_trace.csv>
(rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extra-legs --no-profile).
Takes the LAST 40 occurrences of the step's first kernel (the timed, graph-replayed steps), and for every kernel of a step prints
its mean duration and the mean idle gap in front of it on the device (previous kernel's end -> this kernel's start)."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mel::", "")
first = "plan_enc_kernel"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
idx = idx[-41:]
dur = collections.OrderedDict(); gap = collections.defaultdict(float); cnt = collections.defaultdict(int)
step_total = 0.0
side = ("episode_", "wait_counter")
for a, b in zip(idx[:-1], idx[1:]):
    main = [r for r in rows[a:b] if not any(s in r["Kernel_Name"] for s in side)]
    step_total += (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
    prev_end = None
    for k, r in enumerate(main):
        key = f"{k:02d} {name(r)}"
        dur[key] = dur.get(key, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if prev_end is not None:
            gap[key] += (int(r["Start_Timestamp"]) - prev_end) / 1e3
        prev_end = int(r["End_Timestamp"])
        cnt[key] += 1
    gap["wrap"] += (int(rows[b]["Start_Timestamp"]) - prev_end) / 1e3
n = len(idx) - 1
print(f"{n} steps, {step_total / n:.1f} us from one step's first kernel to the next's")
tot_d = tot_g = 0.0
for key in dur:
    print(f"  {key:<60s} runs {dur[key] / cnt[key]:7.2f} us   idle before it {gap[key] / max(cnt[key], 1):6.2f} us   ({cnt[key]} of {n} steps)")
    tot_d += dur[key] / n; tot_g += gap[key] / n
print(f"  wrap (last kernel's end -> next step's first kernel)  {gap['wrap'] / n:6.2f} us")
print(f"  kernels {tot_d:.1f} us + gaps {tot_g + gap['wrap'] / n:.1f} us")
