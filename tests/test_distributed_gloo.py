"""
CPU, world_size = 2 over gloo: the N > 1 path of the data-parallel front-end.  The forward hot
path has no collective; what must be right is (i) rank-strided utterance ownership identical to
the reference's samplers, (ii) the max-over-ranks timing / summed unit count bench.py reports,
(iii) the all_reduce(sum)/world helper the training path uses (aps/distributed/backend.py:118-129).
"""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from aps_amd import distributed as D
    D.init("torch", "gloo")
    assert D.rank() == rank and D.world_size() == world and D.local_rank() == rank
    mine = D.shard_indices(11, D.rank(), D.world_size())
    dev = torch.device("cpu")
    elapsed = D.reduce_max(1.0 + rank, dev)          # slowest rank defines the step time
    units = D.reduce_sum(32.0 * (rank + 1), dev)     # whole-job unit count
    mean = D.all_reduce(torch.tensor([float(rank)]))  # sum / world
    D.barrier()
    out.put((rank, mine, elapsed, units, mean.item()))


def test_two_rank_sharding_and_reductions():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, e0, u0, m0), (r1, s1, e1, u1, m1) = res
    assert s0 == [0, 2, 4, 6, 8, 10] and s1 == [1, 3, 5, 7, 9]   # indices[rank::world]
    assert sorted(s0 + s1) == list(range(11))                     # every utterance exactly once
    assert e0 == e1 == 2.0 and u0 == u1 == 96.0 and m0 == m1 == 0.5


def test_single_process_defaults():
    from aps_amd import distributed as D
    assert D.rank() == 0 and D.world_size() == 1
    assert D.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert D.reduce_max(3.5, torch.device("cpu")) == 3.5
    t = torch.tensor([2.0])
    assert D.all_reduce(t) is t


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` outside torchrun re-executes itself through torch.distributed.run
    (one rank per GPU): here the launch path alone, with gloo on CPU -- rendezvous on 127.0.0.1,
    barrier + max-over-ranks regions, the summed unit count and the all-reduced `ranks_seen` of
    the JSON line.  (scripts/distributed_train.sh:62-113 of the reference is the launcher it
    replaces.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2",
                          "--selftest-launch", "--steps", "3", "--warmup", "0", "--repeats", "3"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert line["steps"] == 3 and line["ms_per_step_regions"]["repeats"] == 3
    assert line["scaling"] == "weak" and line["value"] > 0
    # whole-job units: both ranks' 32 utterances per step
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 64.0) < 1e-3 * 64.0
