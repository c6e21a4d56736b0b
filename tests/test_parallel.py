"""N > 1 path on CPU: world_size-2 gloo processes exercise env sharding and the one collective of the
path - the flat-gradient all-reduce of the DGN learner."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from melissa_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from melissa_amd.networks import HLDGNNetwork
    from melissa_amd.policy import DQNPolicy
    r, _lr, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    n = 12
    torch.manual_seed(100 + rank)                      # replicas start different on purpose
    net = HLDGNNetwork(5, 64, 2, 2, n, aggregator="max",
                       dueling_param=({"hidden_sizes": [64]}, {"hidden_sizes": [64]}), backend="torch")
    parallel.broadcast_parameters(net, src=0)
    policy = DQNPolicy(net, torch.optim.Adam(net.parameters(), lr=1e-3), estimation_step=1)
    reducer = parallel.FlatGradAllReducer(net)
    # each rank owns its own shard of a global batch of 8 transitions
    g = torch.Generator().manual_seed(7)
    obs = torch.rand(8, 8 * n + 1, generator=g)
    obs[:, -1] = torch.randint(0, n, (8,), generator=g).float()
    act = torch.randint(0, 2, (8,), generator=g)
    ret = torch.randn(8, generator=g)
    lo, hi = parallel.shard_range(8, world, rank)
    for _ in range(3):
        policy.learn(dict(obs=obs[lo:hi], act=act[lo:hi], returns=ret[lo:hi]), grad_hook=reducer)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    np.save(os.path.join(out_dir, f"w{rank}.npy"), flat.numpy())
    total = parallel.all_reduce_sum(float(hi - lo), torch.device("cpu"))
    worst = parallel.all_reduce_max(float(rank), torch.device("cpu"))
    assert total == 8.0 and worst == world - 1
    parallel.barrier()
    torch.distributed.destroy_process_group()


def test_gloo_two_ranks_gradient_allreduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    w0, w1 = np.load(tmp_path / "w0.npy"), np.load(tmp_path / "w1.npy")
    # identical replicas after 3 updates on different shards = gradients were averaged every step
    np.testing.assert_array_equal(w0, w1)

    # and equal to one process training on the whole batch (mean of equal-size shard means)
    from melissa_amd.networks import HLDGNNetwork
    from melissa_amd.policy import DQNPolicy
    n = 12
    torch.manual_seed(100)
    net = HLDGNNetwork(5, 64, 2, 2, n, aggregator="max",
                       dueling_param=({"hidden_sizes": [64]}, {"hidden_sizes": [64]}), backend="torch")
    policy = DQNPolicy(net, torch.optim.Adam(net.parameters(), lr=1e-3), estimation_step=1)
    g = torch.Generator().manual_seed(7)
    obs = torch.rand(8, 8 * n + 1, generator=g)
    obs[:, -1] = torch.randint(0, n, (8,), generator=g).float()
    act = torch.randint(0, 2, (8,), generator=g)
    ret = torch.randn(8, generator=g)
    torch.set_num_threads(1)
    for _ in range(3):
        policy.learn(dict(obs=obs, act=act, returns=ret))
    single = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    np.testing.assert_allclose(w0, single, atol=2e-6, rtol=0)


def test_shard_range_partitions_everything():
    for total, world in [(4096, 8), (1024, 1), (10, 4), (3, 8)]:
        spans = [parallel.shard_range(total, world, r) for r in range(world)]
        covered = [i for lo, hi in spans for i in range(lo, hi)]
        assert covered == list(range(total))


def test_single_process_helpers_are_noops():
    assert parallel.all_reduce_max(3.0, torch.device("cpu")) == 3.0
    assert parallel.all_reduce_sum(3.0, torch.device("cpu")) == 3.0
    parallel.barrier()
