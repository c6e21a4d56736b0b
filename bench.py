#!/usr/bin/env python3
"""bench.py - agent-decisions/s of the Melissa hot path on MI355X.

One "step" = one pass of the hot path over one GPU's shard of envs.  Default (--mode round): one whole env
ROUND per env - ONE L-DGN forward over every (env, active agent) row of the round (the round's agents share
obs_matrix, graph.py:186-188, so encoder / conv1 work is shared) -> argmax / eps-greedy -> ONE env launch
that replays the round's AEC steps and the world step (+ on-device episode reset).  --mode aec: the
reference collector's granularity (multi_agent_collector.py:150-308), one agent decision per env per step.
Per-env trajectories and logits are identical in both modes.
Workload = BASELINE.json's metric config: L-DGN, 50-node graphs, 1024 vectorised envs per GPU
(weak scaling: each rank owns its own 1024 envs, no data-path collective), fp32, dynamic graph (the
reference CLI default, common.py:51), synthetic connected RGGs + seeded random-init weights.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - dominant kernel (by HIP-event time inside the step) vs the fp32 MFMA peak
  cpu_baseline - the CPU oracle (torch restatement + Python env restatement) on the host cores, N=1 only
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0   # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_HBM_GBS = 8000.0            # same guide, HBM3E ~8 TB/s
N_NODES, ENVS_PER_GPU, HIDDEN, HEADS = 50, 1024, 128, 4
N_GRAPHS, RING = 1024, int(os.environ.get("MEL_BENCH_RING", "16"))
MAX_MOVES = int(os.environ.get("MEL_BENCH_MAX_MOVES", "48"))      # movement draws kept per episode (tuning knob)
SETTLE_ROUNDS = int(os.environ.get("MEL_BENCH_SETTLE", "512"))    # untimed rounds after the first reset (build_workload)
HC = HIDDEN * HEADS


def dueling():
    return ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})     # common.py:41-42


GRAPH_ROUNDS = [4]          # rounds per replayed HIP graph (--graph-rounds)


def build_workload(device, rank, envs, n_nodes, model_name, mode, use_graph, streams, seed=9, dtype="f32", supply="stream",
                   prepared_tables=False):
    import torch
    from melissa_amd.collect import DecisionLoop, MultiStreamRoundLoop, RoundLoop
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.networks import DGNRNetwork, HLDGNNetwork, LDGNNetwork
    from melissa_amd.policy import DQNPolicy
    torch.manual_seed(seed)                                   # default --seed 9, common.py:20
    if model_name == "l_dgn":
        net = LDGNNetwork(5, HIDDEN, 2, HEADS, n_nodes, dueling_param=dueling(), device=device, backend="hip")
    elif model_name == "dgn_r":
        net = DGNRNetwork(5, HIDDEN, 2, HEADS, n_nodes, dueling_param=dueling(), device=device, backend="hip")
    else:
        net = HLDGNNetwork(5, HIDDEN, 2, HEADS, n_nodes, aggregator="max", dueling_param=dueling(),
                           device=device, backend="hip")
    net.eval()
    net.set_feature_dtype(dtype)
    net.prepared_tables = bool(prepared_tables)       # legs only: the headline evaluates the node-feature table every call
    # the first 1024 accepted seeds of nx.random_geometric_graph(n, 0.2, seed=s), connected ones only
    graphs = synthetic_graph_pool(n_nodes, N_GRAPHS, first_seed=0)
    make_venv = lambda count, seed: HipGraphVectorEnv(count, n_nodes, graph_pool=graphs, dynamic_graph=True,
                                                      device=device, max_moves=MAX_MOVES, seed=seed,
                                                      construct_like_reference=False)
    policy = DQNPolicy(net)
    base = 1000 + rank * envs
    if mode == "round" and streams > 1:
        loop = MultiStreamRoundLoop(make_venv, policy, envs, streams=streams, seed=base,
                                    eps=0.001, use_graph=use_graph, ring=RING)            # test eps (l_dgn.py:107)
        venv = loop.loops[0].venv
    elif mode == "round":
        venv = make_venv(envs, base)
        # episodes: the device episode stream (every reset draws a new episode, core.py:372-394; ring of 16 slots per env
        # refilled every 7 rounds on a side stream INSIDE the timed region, resets included)
        loop = RoundLoop(venv, policy, seed=base, eps=0.001, use_graph=use_graph, ring=RING,
                         episode_stream=None if supply == "stream" else False, episodes_per_env=12, graph_rounds=GRAPH_ROUNDS[0])
    else:
        venv = make_venv(envs, base)
        loop = DecisionLoop(venv, policy, seed=base, eps=0.001)
    # Untimed settling rounds (set-up, before the W warm-up steps the caller asks for): every env has just been reset, so all
    # of them start in round 0 of an episode and the first rounds of a run evaluate fewer agents per env than the steady
    # state does; after SETTLE_ROUNDS the envs' episode phases are decorrelated and a short timed region (the driver's
    # --steps 20) measures the same loop state as a long one.  The chip too needs a run-up: the same 20 timed steps take
    # 0.185 / 0.183 / 0.179 / 0.171 ms each after 5 / 30 / 100 / 300 warm-up steps on one box (profiles/r03n_warmup_ramp.log;
    # the 2.4 s sustained leg: 0.172) - 512 rounds are ~90 ms of the loop's own work, the state a collector runs in.
    if mode == "round":
        with torch.no_grad():
            loop.run(SETTLE_ROUNDS)
        torch.cuda.synchronize()
    return net, venv, loop


def stage_flops(totals, model="l_dgn", table_rows=0):
    """ALGORITHMIC FLOPs (2*MAC) per launch of each GEMM stage, from the row counts the launch actually
    processed: U1 = conv1 targets, U2 = conv1 sources, R = agent rows (pruned / shared work
    is priced at the pruned / shared count).  DGN-R: key | value projections on the source rows (2 HC wide), query on
    the target rows.  ``table_rows`` > 0: the node-feature table was used - encoder and conv1 projections ran on that
    many tuple rows instead of the U2 / U1 row lists, and are priced at that count."""
    u1, u2, r = totals[0], totals[1], totals[2]
    src = 2 if model == "dgn_r" else 1
    e_rows, l_rows, r_rows = (table_rows,) * 3 if table_rows else (u2, u2, u1)
    return {
        "encoder": e_rows * (2 * 5 * HIDDEN + 2 * HIDDEN * HIDDEN),
        "conv1_lin": 2.0 * (src * l_rows + r_rows) * HC * HIDDEN,  # lin_l + lin_r, one grouped launch
        "conv2_lin": 2.0 * (src * u1 + r) * HC * HC,       # lin_l on U1 rows + lin_r on the agent rows
        "head_hidden": 2.0 * r * ((HIDDEN + 2 * HC) * 256 + 2 * 128 * 128),
    }


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(n_nodes, envs=40, warm=3, reps=10, rep_seconds=1.0):
    """The CPU oracle timed on the host cores: torch fp32 restatement of the forward (all usable cores)
    + step-for-step Python restatement of the env loop, 40 envs (the reference's --training-num).  3 warm-up + 10 timed
    repetitions of a fixed number of collector iterations (calibrated to ~1 s each): median (= ``value``), min, max."""
    import torch
    from oracle import env_oracle as eo
    from oracle import net_oracle as no
    from melissa_amd.env import synthetic_graph_pool
    from melissa_amd.env.episodes import set_to_int
    # torch's CPU ops on these small tensors get slower past a few dozen threads, so the port uses at most
    # 32 of the host cores (the count actually used is what is reported)
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    pool = [eo.GraphSpec(g.pos.copy(), [set_to_int(m) for m in g.one_hop]) for g in synthetic_graph_pool(n_nodes, 8, 0)]
    workers = [eo.OraclePettingZooEnv(eo.OracleGraphEnv(
        n_nodes, graph_pool=pool, dynamic_graph=True,
        np_random=np.random.Generator(np.random.PCG64(np.random.SeedSequence(1000 + k))))) for k in range(envs)]
    sd = no.init_weights("l_dgn", seed=9)
    obs = [w.reset()[0] for w in workers]

    def iterate(count):
        live_total = 0
        with torch.no_grad():
            for _ in range(count):
                batch = np.stack([o["obs"] for o in obs])
                act = no.dqn_act(no.ldgn_forward(sd, batch, n_nodes), np.stack([o["mask"] for o in obs])).numpy()
                for k, w in enumerate(workers):
                    live_total += int(bool(obs[k]["mask"][0]))
                    o, _r, term, _tr, info = w.step(int(act[k]))
                    if term and info.get("explicit_reset"):
                        o, _ = w.reset()
                    obs[k] = o
        return live_total

    t0 = time.perf_counter()
    iterate(2)                                    # thread pool / allocator warm-up + calibration
    per_iter = (time.perf_counter() - t0) / 2
    iters = max(1, int(round(rep_seconds / max(per_iter, 1e-4))))
    rates, decisions = [], 0
    for r in range(warm + reps):
        t0 = time.perf_counter()
        d = iterate(iters)
        dt = time.perf_counter() - t0
        if r >= warm:
            rates.append(d / dt)
            decisions += d
    rates.sort()
    median = rates[len(rates) // 2] if len(rates) % 2 else 0.5 * (rates[len(rates) // 2 - 1] + rates[len(rates) // 2])
    # bs = 1 (the --watch configuration, BASELINE config 0): one env, a few seconds
    one, d1 = workers[0], 0
    obs1, _ = one.reset()
    t1 = time.perf_counter()
    with torch.no_grad():
        while time.perf_counter() - t1 < 3.0:
            a = no.dqn_act(no.ldgn_forward(sd, obs1["obs"][None], n_nodes), np.asarray(obs1["mask"])[None]).numpy()
            d1 += int(bool(obs1["mask"][0]))
            obs1, _r, term, _tr, info = one.step(int(a[0]))
            if term and info.get("explicit_reset"):
                obs1, _ = one.reset()
    bs1 = d1 / (time.perf_counter() - t1)
    return {"value": median, "unit": "agent-decisions/s", "cores": cores, "kind": "port",
            "median": median, "min": rates[0], "max": rates[-1], "repetitions": reps, "warmup_repetitions": warm,
            "cpu_model": cpu_model(), "host_cores": os.cpu_count(), "value_bs1": bs1,
            "sample": f"{reps} timed repetitions (after {warm} warm-up) of {iters} collector iterations over {envs} envs "
                      f"({decisions} live decisions): oracle L-DGN forward (torch CPU fp32, {cores} threads) + Python env "
                      f"restatement, N={n_nodes}, dynamic graph"}


def _env_worker(conn, n_nodes, seeds):
    """One env-worker process of the SubprocVectorEnv-style CPU baseline: owns len(seeds) oracle envs."""
    from oracle import env_oracle as eo
    from melissa_amd.env import synthetic_graph_pool
    from melissa_amd.env.episodes import set_to_int
    pool = [eo.GraphSpec(g.pos.copy(), [set_to_int(m) for m in g.one_hop]) for g in synthetic_graph_pool(n_nodes, 8, 0)]
    envs = [eo.OraclePettingZooEnv(eo.OracleGraphEnv(
        n_nodes, graph_pool=pool, dynamic_graph=True,
        np_random=np.random.Generator(np.random.PCG64(np.random.SeedSequence(s))))) for s in seeds]
    obs = [e.reset()[0] for e in envs]
    conn.send((np.stack([o["obs"] for o in obs]), np.stack([o["mask"] for o in obs])))
    while True:
        act = conn.recv()
        if act is None:
            return
        live = 0
        for k, e in enumerate(envs):
            live += int(bool(obs[k]["mask"][0]))
            o, _r, term, _tr, info = e.step(int(act[k]))
            if term and info.get("explicit_reset"):
                o, _ = e.reset()
            obs[k] = o
        conn.send((np.stack([o["obs"] for o in obs]), np.stack([o["mask"] for o in obs]), live))


def cpu_baseline_subproc(n_nodes, budget_s=8.0, workers=None, envs_per_worker=5):
    """SURVEY.md 8(d) variant (b): one env-worker PROCESS per core group (the reference's SubprocVectorEnv model,
    l_dgn.py:137-146) stepping oracle envs in parallel, one learner process doing the batched oracle forward."""
    import multiprocessing as mp
    import torch
    from oracle import net_oracle as no
    workers = workers or max(1, min(16, (os.cpu_count() or 2) // 2))
    ctx = mp.get_context("spawn")
    conns, procs = [], []
    for w in range(workers):
        a, b = ctx.Pipe()
        p = ctx.Process(target=_env_worker, args=(b, n_nodes, [5000 + w * envs_per_worker + k for k in range(envs_per_worker)]),
                        daemon=True)
        p.start()
        conns.append(a), procs.append(p)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sd = no.init_weights("l_dgn", seed=9)
    first = [c.recv() for c in conns]
    obs, mask = np.concatenate([f[0] for f in first]), np.concatenate([f[1] for f in first])
    decisions, iters = 0, 0
    t0 = time.perf_counter()
    with torch.no_grad():
        while time.perf_counter() - t0 < budget_s:
            act = no.dqn_act(no.ldgn_forward(sd, obs, n_nodes), mask).numpy()
            for w, c in enumerate(conns):
                c.send(act[w * envs_per_worker:(w + 1) * envs_per_worker])
            out = [c.recv() for c in conns]
            obs, mask = np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out])
            decisions += sum(o[2] for o in out)
            iters += 1
    dt = time.perf_counter() - t0
    for c in conns:
        c.send(None)
    for p in procs:
        p.join(timeout=5)
    return {"value": decisions / dt, "workers": workers, "envs": workers * envs_per_worker,
            "sample": f"{iters} iterations, {workers} env-worker processes x {envs_per_worker} envs + one learner ({dt:.1f} s)"}


def timed_run(loop, steps, warmup, device, parallel):
    """W untimed warm-up steps, then EXACTLY ``steps`` steps bracketed by barrier + synchronize on both sides; the time is
    the max over ranks, the counters are summed over ranks (whole-job throughput)."""
    import torch
    loop.run(warmup)
    # the counters before the timed region are COPIED ON THE DEVICE behind the warm-up (one more launch in the queue) and read
    # after it: a host read here leaves the GPU idle for a few hundred microseconds right before t0, and a short timed region
    # (the driver's --steps 20) then starts on a chip that has dropped its clocks
    snap0 = loop.snapshot_counters()
    torch.cuda.synchronize()
    parallel.barrier()
    t0 = time.perf_counter()
    loop.run(steps)
    torch.cuda.synchronize()
    parallel.barrier()
    dt_local = time.perf_counter() - t0
    c0 = loop.counters(snap0)
    c1 = loop.counters()
    local_dec = float(c1["decisions"] - c0["decisions"])
    dt = parallel.all_reduce_max(dt_local, device)
    return {"dt": dt, "dt_local": dt_local, "local_decisions": local_dec,
            "decisions": parallel.all_reduce_sum(local_dec, device),
            "episodes": parallel.all_reduce_sum(float(c1["episodes"] - c0["episodes"]), device),
            "errors": parallel.all_reduce_sum(float(c1["errors"]), device)}


def pmc_traffic():
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE in separate
    passes, gfx950 correction applied by tools/pmc_traffic.py).  The file records the hash of the library sources it was
    taken on; if the kernels have changed since, the bytes no longer belong to the timed launches and are NOT reported."""
    from melissa_amd import build
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")), reverse=True):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if rec.get("source_hash") == build.source_hash():
            return rec.get("per_launch", {}), f"profiles/{name} (rocprofv3 --pmc, same library sources: {rec['source_hash'][:12]})"
        return {}, f"none: profiles/{name} was taken on other library sources ({str(rec.get('source_hash'))[:12]})"
    return {}, "none: no PMC pass committed"


def profiler_average_us(pmc_source, kernel):
    """Average duration of ``kernel`` in the rocprofv3 --kernel-trace --stats summary committed beside the PMC file whose source
    hash matched (tools/refresh_profiles.sh writes both in one go): the profiler's view of the same launch, all launches of the
    profiled command - HIP-graph replays, warm-up and the stage timer's eager pass alike."""
    if not pmc_source.startswith("profiles/"):
        return None
    path = os.path.join(ROOT, pmc_source.split(" ")[0].replace("_pmc_traffic.json", "_round_kernel_stats.csv"))
    needle = kernel.split(" (")[0].replace("mel::", "")
    try:
        for row in open(path):
            if needle in row and row.startswith('"'):
                return float(row.split('",')[1].split(",")[2]) / 1e3          # "Name",Calls,TotalDurationNs,AverageNs,...
    except (OSError, ValueError, IndexError):
        pass
    return None


def stage_profile(args, lib, net, loop, device, rank, decisions_per_step):
    """Per-stage HIP-event timing of the same step in a separate, untimed pass (eager launches of ONE launch chain, each
    launch group bracketed by an event pair on its stream) -> stage_us, the MFMA roofline of the dominant GEMM launch,
    HBM-side rooflines of the gather / env kernels, forward-only and env-only rates."""
    import torch
    from melissa_amd import _lib
    if args.mode == "round" and args.streams > 1:
        torch.cuda.synchronize()
        _n, _v, ploop = build_workload(device, rank, args.envs, args.nodes, args.model, "round", False, 1, dtype=args.dtype)
        ploop.run(args.warmup)
    else:
        ploop = loop
        if args.mode == "round":
            ploop.use_graph = False
    steps = min(args.steps, 200)
    rows_cap = ploop.rows_cap if (args.mode == "round" and args.model != "hl_dgn") else 0
    torch.cuda.synchronize()
    prof = lib.mel_prof_create(steps * 24)
    totals = torch.zeros(steps, 3, dtype=torch.int32, device=device)
    ws = ploop.workspace if args.mode == "round" else net._ws
    lib.mel_prof_attach(prof)
    for k in range(steps):
        ploop.step()
        if args.model != "hl_dgn":
            _lib.check(lib.mel_forward_tap(C.byref(net._weights()), 2, args.envs, args.nodes, rows_cap, ws.data_ptr(),
                                           totals[k].data_ptr(), _lib.current_stream_ptr(device)))
    lib.mel_prof_attach(None)
    torch.cuda.synchronize()
    ms = (C.c_double * _lib.N_STAGES)()
    cnt = (C.c_int64 * _lib.N_STAGES)()
    lib.mel_prof_read(prof, ms, cnt)
    lib.mel_prof_destroy(prof)
    # us per STEP: a stage that brackets several launches per step is their sum; env_reset = the episode stream's refill
    # launches (issued every few rounds on a side stream), averaged over the steps
    stages = {name: (ms[i] / steps * 1e3 if cnt[i] else 0.0) for i, name in enumerate(_lib.STAGE_NAMES)}
    mean_tot = (totals.double().mean(dim=0).cpu().numpy() if args.model != "hl_dgn"
                else (0.0, float(args.envs * args.nodes), float(args.envs)))
    ft = ploop.feature_table() if hasattr(ploop, "feature_table") else {"table_rows": 0, "bad_envs": 0}
    if args.model == "hl_dgn":
        rows_enc = ft["table_rows"] or args.envs * args.nodes
        fl = {"encoder": rows_enc * 34048.0, "conv1_lin": 2.0 * rows_enc * 2 * HC * HIDDEN,
              "head_hidden": 2.0 * args.envs * (HC * 256 + 2 * 128 * 128)}
    else:
        fl = stage_flops(mean_tot, args.model, ft["table_rows"])
    dom = max(fl, key=lambda k: stages.get(k, 0.0))
    pmc, pmc_source = pmc_traffic()
    same = (args.mode == "round" and args.model == "l_dgn" and args.nodes == N_NODES and args.envs == ENVS_PER_GPU
            and args.dtype == "f32a")
    key = {"conv1_lin": "conv1 (lin_l+lin_r)", "conv2_lin": "conv2 (lin_l+lin_r)"}.get(dom)
    traffic = pmc.get(key, {}).get("hbm_bytes_corrected") if same else None
    achieved = fl[dom] / (stages[dom] * 1e-6) / 1e12 if stages[dom] > 0 else 0.0
    # f32s: six bf16 MFMAs per product term set -> the matrix-pipe ceiling for fp32-accurate FLOPs is 2.5 PF / 6
    peak = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "f32s": PEAK_BF16_MFMA_TFLOPS / 6,
            "f32a": PEAK_BF16_MFMA_TFLOPS / 6}[args.dtype]
    tag = {"conv1_lin": 1, "conv2_lin": 2, "head_hidden": 3}.get(dom, 0)
    split_launch = args.dtype == "f32s" or (args.dtype == "f32a" and dom in ("conv2_lin", "head_hidden") and fl[dom] > 2.5e9)
    if args.dtype == "f32a":        # per-launch arithmetic: the ceiling is the one of the pipe the dominant launch runs on
        peak = PEAK_BF16_MFMA_TFLOPS / 6 if split_launch else PEAK_F32_MFMA_TFLOPS
    if args.dtype == "f32a" and split_launch and args.mode == "round" and args.model == "l_dgn":
        # as rocprofv3 summaries name it; conv2's large launches run on the kernel fed from bf16 plane blocks (fwd.hip, MEL_PLANES_FROM)
        planes = dom == "conv2_lin" and "MEL_NO_PLANES_GEMM" not in os.environ
        kname = f"mel::gemm_planes_kernel<{tag}> ({dom})" if planes else f"mel::gemm_split_big_kernel<{tag}> ({dom})"
    elif args.dtype in ("f32", "f32a") and args.mode == "round" and args.model == "l_dgn" and dom != "encoder":
        kname = f"mel::gemm_f32_persistent_kernel<2, 2, 1, 1, 0, {tag}> ({dom})"   # as rocprofv3 summaries name it
    else:
        kname = f"gemm_{args.dtype} ({dom})"
    whole_flops = float(sum(fl.values()))
    step_us = sum(v for k, v in stages.items() if k != "env_reset")
    roofline = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_source": pmc_source if same else "none: PMC passes exist for the default workload only",
                "avg_launch_us": round(stages[dom], 2), "algorithmic_flops_per_launch": fl[dom],
                "profiler_avg_launch_us": None, "frac_by_profiler_avg": None,
                "arithmetic": ("fp32-accurate product on the bf16 matrix pipe: every fp32 operand split exactly into three bf16 pieces, "
                               "six exact partial products per term, fp32 accumulate; `achieved` counts the ALGORITHMIC (fp32) FLOPs, "
                               "`peak` is the bf16 dense MFMA peak 2 500 TF / 6 products; issued bf16 MFMA rate = 6 x achieved"
                               if split_launch else "exact fp32 MFMA (v_mfma_f32_32x32x2_f32)") if args.dtype != "bf16"
                              else "bf16 MFMA, fp32 accumulate",
                "rows_per_launch": {"sum_U1": float(mean_tot[0]), "sum_U2": float(mean_tot[1]), "agent_rows": float(mean_tot[2]),
                                    "feature_table_rows": ft["table_rows"], "envs_with_foreign_features": ft["bad_envs"]},
                "whole_step": {"algorithmic_flops": whole_flops, "stage_sum_us": round(step_us, 2),
                               "achieved": round(whole_flops / (step_us * 1e-6) / 1e12, 3) if step_us > 0 else None,
                               # per-launch precision: the step mixes both matrix pipes, so its algorithmic (fp32) FLOP rate is
                               # quoted against the exact-fp32 MFMA peak - what the step would be bound by without the split launches
                               "peak": PEAK_F32_MFMA_TFLOPS if args.dtype == "f32a" else peak,
                               "frac": round(whole_flops / (step_us * 1e-6) / 1e12 /
                                             (PEAK_F32_MFMA_TFLOPS if args.dtype == "f32a" else peak), 4) if step_us > 0 else None}}

    if same:
        prof_us = profiler_average_us(pmc_source, kname)
        if prof_us:
            roofline["profiler_avg_launch_us"] = round(prof_us, 2)
            roofline["frac_by_profiler_avg"] = round(fl[dom] / (prof_us * 1e-6) / 1e12 / peak, 4)

    # HBM-side rooflines of the non-contraction kernels: COMPULSORY bytes per launch - every row the launch
    # needs read once, every row it produces written once - over the launch's average duration, against the HBM peak.  They
    # are latency / VALU bound (NOTES.md section 5); the fractions say how far from a streaming kernel they are.
    hbm_rooflines = None
    if args.mode == "round" and args.model == "l_dgn":
        esz = 2 if args.dtype == "bf16" else 4
        u1, u2, r = (float(x) for x in mean_tot)
        row = HC * esz
        # x_l rows + x_r rows read (with the node-feature table: at most its rows), h1 + x_1|x_2 written
        tr = ft["table_rows"]
        att1 = ((min(u2, tr) + min(u1, tr)) if tr else (u2 + u1)) * row + u1 * row + r * (HIDDEN + HC) * esz
        att2 = (u1 + r) * row + r * row                                      # x_l2 rows + x_r2 rows read, x_3 written
        env_b = 2.0 * float(lib.mel_env_state_bytes(args.envs, args.nodes))  # every env's state read + written once
        hbm_rooflines = []
        for name, stage, nbytes, key in (("gat_attend_rows_kernel<8, 0, ...> (conv1 attention)", "conv1_att", att1, "conv1 attention"),
                                         ("gat_attend_rows_kernel<8, 2, ...> (conv2 attention)", "conv2_att", att2, "conv2 attention"),
                                         ("env_round_kernel", "env_step", env_b, "env round")):
            us = stages.get(stage, 0.0)
            if us <= 0:
                continue
            gbs = nbytes / (us * 1e-6) / 1e9
            hbm_rooflines.append({"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes_per_launch": int(nbytes),
                                  "avg_launch_us": round(us, 2),
                                  "traffic": pmc.get(key, {}).get("hbm_bytes_corrected") if same else None})
    fwd_us = sum(v for k, v in stages.items() if k not in ("env_step", "env_reset"))
    rows = decisions_per_step if args.mode == "round" else float(args.envs)
    parts = {"forward_only_rows_per_s": rows / (fwd_us * 1e-6) if fwd_us > 0 else None, "forward_us": round(fwd_us, 2),
             "env_only_world_rounds_per_s": (args.envs / (stages["env_step"] * 1e-6)
                                             if args.mode == "round" and stages.get("env_step", 0) > 0 else None),
             "env_only_agent_steps_per_s": (args.envs / (stages["env_step"] * 1e-6)
                                            if args.mode == "aec" and stages.get("env_step", 0) > 0 else None),
             "episode_refill_us_per_step": round(stages.get("env_reset", 0.0), 2)}
    return roofline, stages, hbm_rooflines, parts


def extra_leg(args, device, rank, parallel, note, **over):
    """Another configuration timed exactly like the headline (same barriers, same counters, same steps / warm-up)."""
    kw = dict(envs=args.envs, nodes=args.nodes, model=args.model, mode=args.mode, dtype=args.dtype, streams=1,
              prepared_tables=False)
    kw.update(over)
    import torch
    net, venv, lp = build_workload(device, rank, kw["envs"], kw["nodes"], kw["model"], kw["mode"],
                                   kw["mode"] == "round" and not args.no_graph, kw["streams"], dtype=kw["dtype"],
                                   prepared_tables=kw["prepared_tables"])
    t = timed_run(lp, args.steps, args.warmup, device, parallel)
    out = {"value": t["decisions"] / t["dt"], "unit": "agent-decisions/s", "ms_per_step": t["dt"] / args.steps * 1e3,
           "decisions_per_step": t["decisions"] / args.steps, "env_error_flags": t["errors"],
           "workload": f"{kw['model'].upper().replace('_', '-')} {kw['nodes']}-node, {kw['envs']} envs per GPU, {kw['dtype']}, "
                       f"{'round-batched' if kw['mode'] == 'round' else 'AEC-order'} loop"
                       + (f", {kw['streams']} HIP streams" if kw["streams"] > 1 else ""), "note": note}
    del lp, venv, net
    torch.cuda.empty_cache()
    return out


def learner_leg(args, device, rank, world, parallel, updates=30, rounds_per_update=4, envs=512):
    """BASELINE configs[3]'s per-GPU share WITH its one collective: HL-DGN, 512 envs per GPU, collect 4 rounds then one
    DQN update of batch 32 whose flat fp32 gradient (315 139 elements = 1.26 MB) is summed over the ranks with ONE
    all-reduce (RCCL over xGMI; `allreduce_us` = that collective alone, HIP events, mean of 50)."""
    import torch
    import torch.distributed as dist
    from melissa_amd.collect import RoundLoop
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.networks import HLDGNNetwork
    from melissa_amd.policy import DQNPolicy
    from melissa_amd.replay import DQNLearner, RoundReplay
    torch.manual_seed(9)
    net = HLDGNNetwork(5, HIDDEN, 2, HEADS, args.nodes, aggregator="max", dueling_param=dueling(), device=device)
    parallel.broadcast_parameters(net, src=0)
    policy = DQNPolicy(net, torch.optim.Adam(net.parameters(), lr=1e-3), estimation_step=4, target_update_freq=500)
    venv = HipGraphVectorEnv(envs, args.nodes, graph_pool=synthetic_graph_pool(args.nodes, 64, 0), dynamic_graph=True,
                             device=device, max_moves=48, seed=5000 + rank * envs, construct_like_reference=False)
    replay = RoundReplay(envs, args.nodes, 32, device)
    loop = RoundLoop(venv, policy, seed=5000 + rank * envs, eps=0.1, replay=replay, ring=RING, use_graph=world == 1,
                     graph_rounds=rounds_per_update)        # one rank: an iteration's rounds are one graph replay
    reducer = parallel.FlatGradAllReducer(net)
    learner = DQNLearner(policy, replay, batch_size=32, n_step=4, gamma=0.99, grad_hook=reducer, seed=rank)
    with torch.no_grad():
        loop.run(8)
    learner.step()                                             # warm-up update (lazy init of the autograd kernels, RCCL)
    captured = False
    if world == 1:                    # one rank: the update replayed from HIP graphs (melissa_amd.replay.CapturedUpdate); with
        try:                          # several ranks the eager launches until the captured form has run over RCCL
            learner.capture()
            captured = True
        except Exception as exc:      # noqa: BLE001 - the leg then times the eager update
            learner.captured = None
            print(f"[bench] update not captured: {exc!r}", file=sys.stderr)
    torch.cuda.synchronize()
    parallel.barrier()
    c0 = loop.counters()
    t0 = time.perf_counter()
    for _ in range(updates):
        with torch.no_grad():
            loop.run(rounds_per_update)
        learner.step()
    torch.cuda.synchronize()
    parallel.barrier()
    dt = parallel.all_reduce_max(time.perf_counter() - t0, device)
    dec = parallel.all_reduce_sum(float(loop.counters()["decisions"] - c0["decisions"]), device)
    allreduce_us = None
    if world > 1:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(5):
            dist.all_reduce(reducer.flat)
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(50):
            dist.all_reduce(reducer.flat)
        ev[1].record()
        torch.cuda.synchronize()
        allreduce_us = parallel.all_reduce_max(ev[0].elapsed_time(ev[1]) / 50 * 1e3, device)
    return {"workload": f"HL-DGN {args.nodes}-node, {envs} envs per GPU, fp32: {rounds_per_update} rounds + 1 DQN update (batch 32 per "
                        f"rank, n-step 4) per iteration, flat-gradient all-reduce over {world} rank(s)",
            "updates": updates, "updates_per_s": updates / dt, "value": dec / dt, "unit": "agent-decisions/s",
            "update_replayed_from_hip_graphs": captured,
            "grad_elements": reducer.numel, "grad_bytes": reducer.numel * 4, "allreduce_us": allreduce_us,
            "backend": (dist.get_backend() if world > 1 else None)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU, help="envs per GPU")
    ap.add_argument("--nodes", type=int, default=N_NODES)
    ap.add_argument("--model", default="l_dgn", choices=["l_dgn", "hl_dgn", "dgn_r"])
    ap.add_argument("--mode", default="round", choices=["round", "aec"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--graph-rounds", type=int, default=4,
                    help="rounds (= steps) per replayed HIP graph: a graph launch leaves the device idle ~8 us between replays")
    ap.add_argument("--streams", type=int, default=1,
                    help="round mode: sub-batches of the GPU's envs on separate HIP streams")
    ap.add_argument("--dtype", default="f32a", choices=["f32", "bf16", "f32s", "f32a"],
                    help="f32a (default): fp32 operands, accumulation and results, logits within 1e-4 of the oracle (measured 1e-7); "
                         "each dense projection's arithmetic chosen per launch - the exact-fp32 matrix instruction, or for large "
                         "launches exact bf16 x 3 operand splitting on the bf16 matrix pipe.  f32: every projection on the exact-fp32 "
                         "instruction.  f32s: every projection by operand splitting.  bf16: BASELINE's 'bf16 feature path' (feature "
                         "rows + projection weights bf16, fp32 accumulate / softmax / logits; ~5e-4 on logits)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the extra timed legs of the default run (bf16 feature path, BASELINE configs[1] and [3], "
                         "AEC-order loop, sustained run, learner leg)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--sustained-seconds", type=float, default=2.5)
    ap.add_argument("--episode-supply", default="stream", choices=["stream", "table"],
                    help="stream (default): every reset draws a new episode on the device.  table: round 1's 12 pre-drawn episodes "
                         "per env that wrap (A/B of the stream's cost only; the wrap raises the env's underrun flag)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses cuda:0 and the gloo backend")
    args = ap.parse_args()
    GRAPH_ROUNDS[0] = max(1, args.graph_rounds)

    # --gpus N without an external launcher: this process is still GPU-free (no torch.cuda call, no library load), so it
    # may start the N ranks as fresh children, relay rank 0's JSON line and leave with their exit code.  Under
    # torch.distributed.run (WORLD_SIZE set) it IS a rank; a WORLD_SIZE that contradicts --gpus is refused.
    from melissa_amd import launch
    rc = launch.maybe_spawn(os.path.abspath(__file__), sys.argv[1:], args.gpus,
                            check_devices=not args.rehearse_on_one_gpu)
    if rc is not None:
        raise SystemExit(rc)

    # CPU baseline variant (b) starts child processes: do it BEFORE anything touches the GPU (no fork / exec from a
    # process with an initialised HIP runtime)
    subproc_baseline = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        try:
            subproc_baseline = cpu_baseline_subproc(args.nodes)
        except Exception as exc:              # the baseline is a report, never a reason to lose the bench line
            subproc_baseline = {"error": repr(exc)}

    import torch
    from melissa_amd import _lib, parallel
    if not args.rehearse_on_one_gpu:
        launch.check_rank_device()            # a rank whose LOCAL_RANK names a GPU it cannot see leaves with code 2
    rank, local_rank, world = parallel.init_distributed("gloo" if args.rehearse_on_one_gpu else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the hot path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    net, venv, loop = build_workload(device, rank, args.envs, args.nodes, args.model, args.mode, not args.no_graph,
                                     args.streams, dtype=args.dtype, supply=args.episode_supply)
    lib = _lib.load()

    t = timed_run(loop, args.steps, args.warmup, device, parallel)
    dt, decisions, episodes, errors = t["dt"], t["decisions"], t["episodes"], t["errors"]
    # per-rank rates (each rank's own decisions over its own wall time), gathered for the line
    per_rank = [0.0] * world
    per_rank[rank] = t["local_decisions"] / t["dt_local"]
    per_rank = [parallel.all_reduce_sum(v, device) for v in per_rank]
    supply = loop.supply.describe() if hasattr(loop, "supply") else loop.loops[0].supply.describe()

    # ---- sustained leg: the same loop for >= ~2.5 s of wall time (the timed region above is a few ms at the driver's
    # --steps 20, too short for a utilisation sampler to see the GPU busy)
    sustained = None
    if not args.no_extra_legs and args.sustained_seconds > 0:
        n_sus = max(args.steps, int(args.sustained_seconds / max(dt / args.steps, 1e-6)))
        ts = timed_run(loop, n_sus, 0, device, parallel)
        sustained = {"steps": n_sus, "seconds": ts["dt"], "value": ts["decisions"] / ts["dt"], "unit": "agent-decisions/s",
                     "ms_per_step": ts["dt"] / n_sus * 1e3, "episodes_finished": ts["episodes"], "env_error_flags": ts["errors"]}

    roofline = stages = hbm_rooflines = parts = None
    if not args.no_profile and rank == 0:
        roofline, stages, hbm_rooflines, parts = stage_profile(args, lib, net, loop, device, rank,
                                                               decisions / args.steps / world)
    parallel.barrier()

    # ---- extra legs: other BASELINE configurations, timed like the headline -----------------------------------------------
    legs = None
    default_cfg = (args.dtype == "f32a" and args.mode == "round" and args.streams == 1 and args.model == "l_dgn"
                   and args.nodes == N_NODES)
    if default_cfg and not args.no_extra_legs:
        del loop, venv
        torch.cuda.empty_cache()
        specs = {
            "bf16_feature_path": dict(note="BASELINE configs[2] as worded: feature rows + projection weights bf16, fp32 accumulate / "
                                           "softmax / logits; logits within 5e-4 of the fp32 path", dtype="bf16"),
            "config1_ldgn_n20_256envs": dict(note="BASELINE configs[1]: L-DGN 20-node, 256 vectorised envs, fp32", nodes=20, envs=256),
            "config3_hldgn_512envs_per_gpu": dict(note="BASELINE configs[3]'s per-GPU share: HL-DGN 50-node, 4096 envs over 8 GPUs = 512 "
                                                       "per GPU (collect path; its collective is in learner_leg)", model="hl_dgn", envs=512),
            "ldgn_n100_1024envs": dict(note="the reference CLI's third graph size (--n-agents 100, common.py:49; in no BASELINE config): "
                                            "two-word node sets, two nodes per wavefront lane - same kernels, same loop", nodes=100),
            "aec_order_loop": dict(note="the reference collector's granularity (multi_agent_collector.py:150-308): one agent decision "
                                        "per env per step", mode="aec"),
            "two_streams": dict(note="the headline workload as two half-batches of 512 envs on two HIP streams, each replaying its own "
                                     "HIP graph (the latency-bound launches of one half fill the gaps of the other); not the headline "
                                     "because overlapping launches cannot be priced per kernel", streams=2),
            "prepared_feature_tables": dict(note="the headline workload with the node-feature table (encoder + conv1 projections of the "
                                                 "2 000 feature tuples: a function of the weights only) prepared ONCE per weight "
                                                 "version (mel_prepare_feature_tables) instead of evaluated inside every step as the "
                                                 "headline does; bit-identical logits", prepared_tables=True),
            "exact_fp32_mfma_only": dict(note="every projection on the exact-fp32 matrix instruction (MEL_PREC_F32; the headline "
                                              "precision mode MEL_PREC_F32_AUTO sends conv2 and the heads' first layer to the split-bf16 "
                                              "kernels at this size): round 2's earlier headline", dtype="f32"),
            "f32_via_split_bf16_mfma": dict(note="fp32-accurate projections on the bf16 matrix cores (every operand split exactly into three "
                                                 "bf16 pieces, six partial products; logits 3-7e-8 from the oracle like the native path)",
                                            dtype="f32s"),
        }
        if world > 1:
            # N > 1: only the configuration BASELINE quotes on several GPUs (configs[3]) besides the headline - every leg is a
            # chain of collectives (barriers, reductions) that all ranks must walk in step, so the fewer the better
            specs = {k: v for k, v in specs.items() if k == "config3_hldgn_512envs_per_gpu"}
        legs = {}
        for name, spec in specs.items():
            try:
                legs[name] = extra_leg(args, device, rank, parallel, **spec)
            except Exception as exc:          # an extra leg must never cost the headline line (one rank: no barrier to desync)
                if world > 1:
                    raise
                legs[name] = {"error": repr(exc)}
        try:
            legs["learner_leg"] = learner_leg(args, device, rank, world, parallel)
        except Exception as exc:
            if world > 1:
                raise
            legs["learner_leg"] = {"error": repr(exc)}

    parallel.barrier()
    if rank != 0:
        return
    value = decisions / dt
    prec = dict(f32="fp32", bf16="bf16 feature path", f32s="fp32 via split-bf16 MFMA",
                f32a="fp32, arithmetic per launch (large projections via split-bf16 MFMA, the rest exact fp32 MFMA)")[args.dtype]
    line = {
        "metric": "env-steps/s (agent-decisions/s) L-DGN 50-node" if args.model == "l_dgn" and args.nodes == 50
                  else f"env-steps/s (agent-decisions/s) {args.model} {args.nodes}-node",
        "value": value, "unit": "agent-decisions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # f32a: fp32 operands / accumulation / results; the large launches multiply by exact bf16 x 3 operand splitting (see
        # precision_mode) - named as such, not as plain "f32"
        "dtype": "f32 (f32a)" if args.dtype == "f32a" else args.dtype, "data": "synthetic",
        "precision_mode": {"f32a": "MEL_PREC_F32_AUTO: fp32 operands, fp32 results, fp32 accumulation everywhere (logits within 1e-4 of the "
                                   "oracle, measured 1e-7); the arithmetic of each dense projection is chosen per launch by its size - the "
                                   "exact-fp32 matrix instruction, or for the large launches (here conv2 and the heads' first layer) exact "
                                   "bf16 x 3 operand splitting on the bf16 matrix pipe with six exact partial products per term",
                           "f32": "MEL_PREC_F32: every projection on the exact-fp32 matrix instruction",
                           "f32s": "MEL_PREC_F32_SPLIT: every projection by exact bf16 x 3 operand splitting on the bf16 matrix pipe",
                           "bf16": "MEL_PREC_BF16: bf16 feature rows and projection weights, fp32 accumulate"}[args.dtype],
        "config": {"workload": f"{args.model.upper().replace('_', '-')} {args.nodes}-node, {args.envs} vectorised envs per GPU, {prec}, "
                               f"dynamic graph, eps=0.001, "
                               + ("round-batched loop (one env round per step)" if args.mode == "round"
                                  else "AEC-order loop (one agent decision per env per step)")
                               + f"; graphs: the first {N_GRAPHS} connected nx.random_geometric_graph(n, 0.2, seed=s); episodes: "
                               + (f"device episode stream - every reset draws a NEW episode (graph, source, interested set, movement "
                                  f"seed) with the reference's RNG protocol, ring of {supply.get('ring')} slots per env refilled every "
                                  f"{supply.get('refill_every')} steps on a side stream inside the timed region, reset snapshots rebuilt "
                                  f"by the refill (no episode is ever replayed); {SETTLE_ROUNDS} untimed settling rounds after the first reset, "
                                  f"before the warm-up" if supply["mode"] == "device stream"
                                  else f"static table of {supply.get('episodes_per_env')} pre-drawn episodes per env"),
                   "loop": args.mode, "hip_graph": bool(args.mode == "round" and not args.no_graph),
                   "rounds_per_graph_replay": (GRAPH_ROUNDS[0] if args.mode == "round" and not args.no_graph and args.streams == 1 else 1),
                   "streams": args.streams if args.mode == "round" else 1,
                   "envs_per_gpu": args.envs, "n_nodes": args.nodes, "global_envs": args.envs * world,
                   "parallelism": f"env-shard x{world} (no data-path collective)", "episode_supply": supply,
                   "decisions_per_step": decisions / args.steps, "live_decisions": decisions, "episodes_finished": episodes,
                   "env_error_flags": errors},
        "rccl_ranks": world if (world > 1 and not args.rehearse_on_one_gpu) else 0,
        "per_rank_value": per_rank,
        "roofline": roofline,
        "sustained": sustained,
        "legs": legs,
        "stage_us": {k: round(v, 2) for k, v in (stages or {}).items() if v > 0},
        "parts": parts,
        "roofline_hbm": hbm_rooflines,
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.nodes)
        line["cpu_baseline"]["subproc"] = subproc_baseline       # variant (b): env-worker processes + one learner
    print(json.dumps(line))


if __name__ == "__main__":
    main()
