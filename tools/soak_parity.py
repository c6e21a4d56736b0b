"""Soak: the round-loop-vs-oracle parity test (tests/test_gpu_round.py) at more sizes than the suite runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_round as t
for n, dyn in ((50, True), (33, True), (64, True), (7, False), (20, False), (12, True), (50, False), (41, True), (64, False)):
    t0 = time.time()
    t.test_round_loop_matches_oracle(n, dyn)
    print(f"n={n} dynamic={dyn}: ok ({time.time() - t0:.1f} s)", flush=True)
for scripted in (None, (0.3, "simple_broadcast"), (0.4, "broadcast_if_any_interested"), (0.5, "silent"), (0.2, "simple_broadcast")):
    t0 = time.time()
    t.test_hldgn_round_loop_matches_oracle(scripted)
    print(f"hl_dgn scripted={scripted}: ok ({time.time() - t0:.1f} s)", flush=True)
