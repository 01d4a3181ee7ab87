"""CPU, world_size 2, gloo: the draw-sharded evaluator gathers per-draw (logp, status) into
rank order.  The local evaluation is a stand-in closure (the HIP engine needs a GPU); what is
under test is the N > 1 path of geconpy_amd.engine.ShardedLogpEvaluator: contiguous shards,
ragged tail, rank-ordered all-gather, bit-exact draw indexing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _value_of_draw(i):
    # deterministic per-draw value with no structure a wrong ordering could preserve
    return np.sin(1.0 + 0.37 * np.asarray(i, dtype=np.float64)) * 1e3


def _worker(rank, world, port, global_batch, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from geconpy_amd.engine import ShardedLogpEvaluator

    def local_eval(lo, hi):
        idx = np.arange(lo, hi)
        logp = torch.from_numpy(_value_of_draw(idx))
        status = torch.from_numpy((idx % 5 == 0).astype(np.int32))
        return logp, status

    ev = ShardedLogpEvaluator(global_batch, local_eval, torch.device("cpu"))
    for _ in range(2):  # buffers are reused across steps
        logp, status = ev.step()
    q.put((rank, ev.lo, ev.hi, logp.numpy().copy(), status.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [64, 37])
def test_sharded_gather_world2(global_batch):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, global_batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = _value_of_draw(np.arange(global_batch))
    exp_status = (np.arange(global_batch) % 5 == 0).astype(np.int32)
    bounds = sorted((lo, hi) for _, lo, hi, _, _ in results)
    assert bounds[0][0] == 0 and bounds[-1][1] == global_batch and bounds[0][1] == bounds[1][0]
    for rank, lo, hi, logp, status in results:
        assert logp.shape == (global_batch,)
        assert np.array_equal(logp, expected)  # bit-exact, rank-ordered, on every rank
        assert np.array_equal(status, exp_status)


def test_single_process_passthrough():
    from geconpy_amd.engine import ShardedLogpEvaluator

    ev = ShardedLogpEvaluator(10, lambda lo, hi: (torch.arange(lo, hi, dtype=torch.float64), torch.zeros(hi - lo, dtype=torch.int32)),
                              torch.device("cpu"))
    logp, status = ev.step()
    assert (ev.lo, ev.hi) == (0, 10) and torch.equal(logp, torch.arange(10, dtype=torch.float64))
