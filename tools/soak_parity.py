"""Soak: the round-loop-vs-oracle parity tests (tests/test_gpu_round.py) at more sizes than the suite runs, with both episode
supplies (host-drawn table / device episode stream).  Prints one line per case; run it on the GPU box and keep the output
under profiles/ (python tools/soak_parity.py [part] | tee gpurun_out/soak_parity.log).  The CPU oracle dominates the run
time (minutes per case beyond 64 nodes), so the cases come in parts that each fit one gpurun call:
  ldgn       L-DGN, graphs of up to 64 nodes (one-word node sets)
  ldgn_wide  L-DGN, 65 .. 128 nodes (two-word node sets)
  hldgn      HL-DGN with and without scripted agents, 20 .. 100 nodes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_round as t

PARTS = {
    "ldgn": [(50, True, "table"), (33, True, "stream"), (64, True, "stream"), (7, False, "table"), (20, False, "stream"),
             (12, True, "table"), (50, False, "stream"), (41, True, "stream"), (64, False, "table"), (50, True, "stream")],
    "ldgn_wide": [(100, True, "stream"), (65, True, "table"), (97, True, "stream"), (100, False, "table"), (128, False, "table")],
}
part = sys.argv[1] if len(sys.argv) > 1 else "all"
for name, cases in PARTS.items():
    if part not in ("all", name):
        continue
    for n, dyn, supply in cases:
        t0 = time.time()
        t.round_loop_vs_oracle(n, dyn, supply)
        print(f"l_dgn round loop n={n} dynamic={dyn} episodes={supply}: bit-exact env state + logits within 1e-4 of the oracle "
              f"({time.time() - t0:.1f} s)", flush=True)
if part in ("all", "hldgn"):
    for n in (20, 50, 37, 100):
        for scripted in (None, (0.3, "simple_broadcast"), (0.4, "broadcast_if_any_interested"), (0.5, "silent"), (0.2, "simple_broadcast")):
            t0 = time.time()
            t.test_hldgn_round_loop_matches_oracle(scripted, n)
            print(f"hl_dgn round loop n={n} scripted={scripted}: ok ({time.time() - t0:.1f} s)", flush=True)
print(f"soak ok ({part})")
