"""
Multi-GPU plumbing on hardware -- skipped below two visible devices (the 1-GPU test box), so that the
first scaling curve the driver takes on an 8-GPU node cannot fail on the launch path:
  * `python bench.py --gpus 2` over RCCL (self-launch through torch.distributed.run, rank -> device
    binding, barriers, max-over-ranks timing): both ranks seen, different input shards per rank, whole-job
    utterances = 2 x the per-rank batch, per-rank step time within 10 % of the one-GPU run;
  * DistributedDataParallel over the HIP modules with the nccl (= RCCL) backend, one rank per device:
    identical averaged gradients on both ranks (aps/trainer/ddp.py:107-165).
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def two_gpus():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")


def _bench(gpus):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "4",
                          "--warmup", "2", "--repeats", "3", "--group", "1", "--batches", "2",
                          "--no-cpu-baseline", "--no-baseline-batch"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout  # rank 0 only
    return json.loads(lines[0])


def test_bench_two_ranks_over_rccl(two_gpus):
    one, two = _bench(1), _bench(2)
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["scaling"] == "weak"
    assert two["inputs_distinct_across_ranks"] is True and one["inputs_distinct_across_ranks"] is None
    assert two["config"]["global_batch"] == 2 * one["config"]["global_batch"]
    # whole-job units: both ranks' 32 utterances per step
    assert abs(two["value"] * two["ms_per_step"] * 1e-3 - 64.0) < 1e-3 * 64.0
    ratio = two["ms_per_step"] / one["ms_per_step"]
    print(f"[multi] ms per step: 1 GPU {one['ms_per_step']:.3f}, 2 GPUs {two['ms_per_step']:.3f} "
          f"(max over ranks): x {ratio:.3f}")
    assert ratio < 1.10, "no collective on the data path: a rank must not slow down next to another"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from aps_amd import distributed as D
    from tests.test_gpu_train import joint_inputs, small_joint
    device = torch.device("cuda", rank)
    torch.cuda.set_device(device)
    D.init("torch", "nccl")  # RCCL over xGMI: one rank per device
    net = small_joint(seed=46).eval().to(device)
    ddp = DDP(net, device_ids=[rank], **D.ddp_kwargs())
    wav, lens, g = joint_inputs(seed=100 + rank)  # a different shard per rank
    enc_out, enc_ctc, _ = ddp(wav.to(device), lens.to(device))
    (enc_out.square().mean() + enc_ctc.square().mean()).backward()
    grads = {n: p.grad.detach().cpu().numpy() for n, p in net.named_parameters() if p.grad is not None}
    dist.barrier()
    out.put((rank, grads))
    dist.destroy_process_group()


def test_ddp_gradient_all_reduce_over_rccl(two_gpus):
    import torch.multiprocessing as mp
    from tests.test_gpu_train import joint_inputs, small_joint
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    device = torch.device("cuda:0")
    single = []
    for rank in range(world):
        net = small_joint(seed=46).eval().to(device)
        wav, lens, g = joint_inputs(seed=100 + rank)
        enc_out, enc_ctc, _ = net(wav.to(device), lens.to(device))
        (enc_out.square().mean() + enc_ctc.square().mean()).backward()
        single.append({n: p.grad.detach().cpu() for n, p in net.named_parameters() if p.grad is not None})
    assert res[0].keys() == res[1].keys() == single[0].keys()
    for name in res[0]:
        a, b = torch.from_numpy(res[0][name]), torch.from_numpy(res[1][name])
        assert torch.equal(a, b), f"{name}: ranks disagree"
        mean = 0.5 * (single[0][name] + single[1][name])
        scale = max(single[0][name].abs().max().item(), single[1][name].abs().max().item(), 1e-30)
        assert (a - mean).abs().max().item() / scale <= 1e-5, name


def test_pinned_stager_overlaps_the_copy_with_compute(device):
    """round 5 (advisor): PinnedStager's pipelined form really runs the H2D copy of batch k + 1 beside the kernels
    of batch k -- the copy waits for the release event of ITS slot's last reader, not for the head of the compute
    stream.  A 128 MB transfer (~2 ms over PCIe) next to a ~4 ms spin on the compute stream: staged ahead, the
    pair takes about as long as the longer of the two, not their sum; the data arrives intact; the one-call
    `put` stays correct."""
    from aps_amd import distributed as D
    shape = (32, 4, 250000)   # 128 MB of float32
    st = D.PinnedStager(shape, device=device, depth=2)
    a = torch.arange(32 * 4 * 250000, dtype=torch.float32).view(shape)
    b = -a

    def timed(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    probe = min(timed(lambda: torch.cuda._sleep(1_000_000)) for _ in range(3))   # ms per 1e6 "cycles"
    spin = int(1_000_000 * 4.0 / max(probe, 1e-3))   # ~4 ms on the compute stream
    t_spin = min(timed(lambda: torch.cuda._sleep(spin)) for _ in range(3))
    slot = st.stage(a)
    x = st.use(slot)
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), a)

    def copy_alone():
        s2 = st.stage(b)
        st.use(s2)
        st.release(s2)
    t_copy = min(timed(copy_alone) for _ in range(3))

    def pipelined():
        ahead = st.stage(a)            # queued first: runs beside the spin
        torch.cuda._sleep(spin)        # "the kernels of batch k" on the compute stream
        st.release(slot)
        y = st.use(ahead)
        st.release(ahead)
        return y
    t_both = min(timed(pipelined) for _ in range(3))
    print(f"[stager] spin {t_spin:.2f} ms, copy {t_copy:.2f} ms, staged ahead {t_both:.2f} ms")
    assert t_copy > 0.3 and t_spin > 1.0
    assert t_both < t_spin + 0.5 * t_copy, (t_spin, t_copy, t_both)
    y = st.put(b)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), b)
