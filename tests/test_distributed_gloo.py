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


def _run_ranks(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_two_rank_sharding_and_reductions():
    (r0, s0, e0, u0, m0), (r1, s1, e1, u1, m1) = _run_ranks(2)
    assert s0 == [0, 2, 4, 6, 8, 10] and s1 == [1, 3, 5, 7, 9]   # indices[rank::world]
    assert sorted(s0 + s1) == list(range(11))                     # every utterance exactly once
    assert e0 == e1 == 2.0 and u0 == u1 == 96.0 and m0 == m1 == 0.5


def test_eight_rank_sharding_and_reductions():
    """the node the metric is quoted on: 8 ranks (one per MI355X), here over gloo on the CPU"""
    res = _run_ranks(8)
    shards = [r[1] for r in res]
    assert shards[3] == [3] and shards[0] == [0, 8] and shards[2] == [2, 10]
    assert sorted(i for s in shards for i in s) == list(range(11))
    for rank, _, elapsed, units, mean in res:
        assert elapsed == 8.0                    # the slowest rank (1.0 + 7) defines the step
        assert units == 32.0 * 36                # 32 * (1 + ... + 8)
        assert mean == 3.5


def test_rank_cores_and_pinned_stager():
    """host side of 1 -> 8 scaling: every rank gets a contiguous, disjoint, equal share of the node's
    cores; the stager hands batches over through its ring of buffers (page-locked on a GPU box)"""
    from aps_amd import distributed as D
    cores = list(range(256))
    shares = [D.rank_cores(r, 8, cores) for r in range(8)]
    assert all(len(s) == 32 for s in shares) and shares[3] == list(range(96, 128))
    assert sorted(c for s in shares for c in s) == cores
    assert D.rank_cores(0, 1, cores) == cores and D.rank_cores(2, 8, [0, 1, 2]) == [0, 1, 2]
    assert D.bind_rank_to_cores() == []          # a single rank: nothing to do
    st = D.PinnedStager((2, 3), depth=2)
    a, b, c = torch.ones(2, 3), 2 * torch.ones(2, 3), 3 * torch.ones(2, 3)
    ra = st.put(a)
    assert torch.equal(ra, a)
    rb = st.put(b)
    assert torch.equal(rb, b) and torch.equal(ra, a)      # two buffers: the first is still intact
    rc = st.put(c)
    assert torch.equal(rc, c) and rc.data_ptr() == ra.data_ptr()
    # the pipelined form: stage ahead, use, release
    st = D.PinnedStager((2, 3), depth=2)
    slot = st.stage(a)
    for nxt, want in ((b, a), (c, b), (a, c)):
        ahead = st.stage(nxt)
        assert torch.equal(st.use(slot), want)
        st.release(slot)
        slot = ahead
    kw = D.ddp_kwargs()
    assert kw["gradient_as_bucket_view"] and kw["static_graph"] and kw["bucket_cap_mb"] == 32


def test_single_process_defaults():
    from aps_amd import distributed as D
    assert D.rank() == 0 and D.world_size() == 1
    assert D.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert D.reduce_max(3.5, torch.device("cpu")) == 3.5
    t = torch.tensor([2.0])
    assert D.all_reduce(t) is t


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 8])
def test_bench_self_launches_its_ranks(world):
    """`python bench.py --gpus N` outside torchrun re-executes itself through torch.distributed.run
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
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world),
                          "--selftest-launch", "--steps", "3", "--warmup", "0", "--repeats", "3"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["ranks_seen"] == world
    assert line["steps"] == 3 and line["ms_per_step_regions"]["repeats"] == 3
    assert line["scaling"] == "weak" and line["value"] > 0
    # whole-job units: every rank's 32 utterances per step
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 32.0 * world) < 1e-3 * 32.0 * world
