"""
GraphReplicas (aps_amd/replicas.py): R captured copies of one step replayed on R streams give the
eager result, replay after replay, with an LSTM (memory-synchronised persistent grid) in the step.
"""
import pytest
import torch as th

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("replicas", [1, 2])
def test_replicas_match_eager(replicas, device):
    from aps_amd import nn_ops
    from aps_amd.replicas import GraphReplicas, concurrent_launches
    th.manual_seed(3)
    dev = device
    rnn = th.nn.LSTM(128, 128, num_layers=2, batch_first=True).eval()
    proj = th.nn.Linear(128, 96).eval()
    x_cpu = th.randn(8, 20, 128)
    with th.no_grad():
        want = proj(rnn(x_cpu)[0])
    rnn_d, proj_d, x = rnn.to(dev), proj.to(dev), x_cpu.to(dev)

    def step():
        return nn_ops.linear(nn_ops.lstm_forward(rnn_d, x), proj_d.weight, proj_d.bias)

    with th.no_grad():
        with concurrent_launches(replicas):
            eager = step()
        reps = GraphReplicas(step, replicas=replicas)
        for _ in range(3 * replicas):
            reps.submit()
        reps.synchronize()
        for out in reps.outputs:
            assert th.equal(out, eager)
        # new input through the same static tensor: every replica sees it on its next replay
        x.mul_(0.5)
        with concurrent_launches(replicas):
            eager2 = step()
        for _ in range(replicas):
            index, _ = reps.submit()
            assert th.equal(reps.wait(index), eager2)
    scale = want.abs().max().item()
    assert (eager.cpu() - want).abs().max().item() <= 1e-4 * scale


def test_replicas_reject_zero():
    from aps_amd.replicas import GraphReplicas
    with pytest.raises(ValueError):
        GraphReplicas(lambda: None, replicas=0)


@pytest.mark.parametrize("share", [2, 4])
@pytest.mark.parametrize("N,layers,bidir", [(32, 2, False), (64, 1, True), (48, 3, False)])
def test_lstm_sized_for_a_share_of_the_chip(share, N, layers, bidir, device):
    """grids sized for 1 / share of the resident slots (or chunked when no shape fits) still give
    nn.LSTM's numbers"""
    from aps_amd import nn_ops
    from aps_amd.replicas import concurrent_launches
    th.manual_seed(11 + N + layers)
    rnn = th.nn.LSTM(256, 512, num_layers=layers, batch_first=True, bidirectional=bidir).eval()
    x = th.randn(N, 24, 256)
    with th.no_grad():
        want = rnn(x)[0]
        dev = device
        rnn_d = rnn.to(dev)
        with concurrent_launches(share):
            got = nn_ops.lstm_forward(rnn_d, x.to(dev)).cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 1e-4 * want.abs().max().item()


@pytest.mark.parametrize("workers,share,front", [(1, 1, "head"), (3, 2, "head"), (2, 1, "worker"), (3, 2, "worker")])
def test_pipelined_replicas_match_eager(workers, share, front, device):
    """PipelinedReplicas: the step cut at its persistent LSTM-stack launch into three hipGraphs (worker stream ->
    the one LSTM stream -> worker stream) gives the eager step's bits for every resident batch, replay after
    replay, and sees new inputs through the static tensors"""
    from aps_amd import nn_ops
    from aps_amd.replicas import PipelinedReplicas, concurrent_launches
    th.manual_seed(5)
    dev = device
    rnn = th.nn.LSTM(128, 128, num_layers=2, batch_first=True).eval()
    pre = th.nn.Linear(96, 128).eval()
    proj = th.nn.Linear(128, 96).eval()
    xs_cpu = [th.randn(8, 20, 96) for _ in range(4)]
    with th.no_grad():
        want = [proj(rnn(pre(x))[0]) for x in xs_cpu]
    rnn_d, pre_d, proj_d = rnn.to(dev), pre.to(dev), proj.to(dev)
    xs = [x.to(dev) for x in xs_cpu]

    def step(x):
        h = nn_ops.linear(x, pre_d.weight, pre_d.bias)
        return nn_ops.linear(nn_ops.lstm_forward(rnn_d, h), proj_d.weight, proj_d.bias)

    with th.no_grad():
        before = (nn_ops.lstm_share(), nn_ops.STREAMS_IN_FLIGHT)
        reps = PipelinedReplicas([lambda x=x: step(x) for x in xs], workers=workers, lstm_share=share, front=front)
        assert reps.stages == 3 and len(reps) == 4
        assert [on_lstm for _, on_lstm in reps.pipelines[0]] == [False, True, False]
        assert reps.kinds[0] == ["a", "l", "b"]   # (no "enhance_end" in this step: nothing stays behind on the head stream)
        assert nn_ops.lstm_share() == share and nn_ops.STREAMS_IN_FLIGHT == workers + 1
        for _ in range(5 * len(reps)):
            reps.submit()
        reps.synchronize()
        reps.check_outputs(reps.eager_outputs, "after 20 submissions")
        for out, w in zip(reps.outputs, want):
            assert (out.cpu() - w).abs().max().item() <= 1e-4 * w.abs().max().item()
        # new input through the static tensors; the eager step under the same library state
        for x in xs:
            x.mul_(0.5)
        eager2 = [step(x) for x in xs]
        for _ in range(len(reps)):
            index, _ = reps.submit()
            assert th.equal(reps.wait(index), eager2[index])
        reps.close()
        assert (nn_ops.lstm_share(), nn_ops.STREAMS_IN_FLIGHT) == before
        reps.close()  # idempotent
        assert (nn_ops.lstm_share(), nn_ops.STREAMS_IN_FLIGHT) == before


def test_pipelined_replicas_single_stage_without_a_persistent_launch(device):
    """a step with no LSTM stack is one stage: the class degenerates to graphs round-robin on its worker streams"""
    from aps_amd import nn_ops
    from aps_amd.replicas import PipelinedReplicas
    th.manual_seed(6)
    lin = th.nn.Linear(64, 48).eval().to(device)
    xs = [th.randn(16, 64, device=device) for _ in range(3)]
    with th.no_grad():
        reps = PipelinedReplicas([lambda x=x: nn_ops.linear(x, lin.weight, lin.bias) for x in xs], workers=2)
        assert reps.stages == 1
        for _ in range(6):
            index, _ = reps.submit()
        reps.synchronize()
        for x, out in zip(xs, reps.outputs):
            assert th.equal(out, nn_ops.linear(x, lin.weight, lin.bias))
        reps.close()


@pytest.mark.parametrize("layers,bidir", [(1, False), (2, True), (1, True)])
def test_pipelined_replicas_move_every_persistent_launch_to_the_lstm_stream(layers, bidir, device):
    """single-layer and bidirectional recurrences launch aps_lstm_layer (one persistent launch per layer),
    not the stack kernel; those launches are stages of their own too -- captured on the ONE stream that runs the
    persistent launches of all batches one after the other -- so `workers` batches never bring three
    memory-synchronised grids (each sized for 1 / lstm_share of the chip) onto the chip at once"""
    from aps_amd import nn_ops
    from aps_amd.replicas import PipelinedReplicas
    th.manual_seed(21 + layers)
    rnn = th.nn.LSTM(128, 128, num_layers=layers, batch_first=True, bidirectional=bidir).eval()
    proj = th.nn.Linear(256 if bidir else 128, 64).eval()
    xs_cpu = [th.randn(8, 20, 128) for _ in range(4)]
    with th.no_grad():
        want = [proj(rnn(x)[0]) for x in xs_cpu]
    rnn_d, proj_d = rnn.to(device), proj.to(device)
    xs = [x.to(device) for x in xs_cpu]

    def step(x):
        return nn_ops.linear(nn_ops.lstm_forward(rnn_d, x), proj_d.weight, proj_d.bias)

    with th.no_grad():
        reps = PipelinedReplicas([lambda x=x: step(x) for x in xs], workers=3, lstm_share=2)
        kinds = reps.kinds[0]
        assert kinds.count("l") == layers, kinds          # one persistent launch per layer (both directions in it)
        assert kinds[0] == "a" and kinds[-1] == "b", kinds
        for k, (_, on_lstm) in zip(kinds, reps.pipelines[0]):
            assert on_lstm == (k == "l")
        for _ in range(6 * len(reps)):
            reps.submit()
        reps.synchronize()                                # (raises on a hand-off timeout)
        reps.check_outputs(reps.eager_outputs, "after 24 submissions")
        for out, w in zip(reps.outputs, want):
            assert (out.cpu() - w).abs().max().item() <= 1e-4 * w.abs().max().item()
        reps.close()


@pytest.mark.parametrize("order", ["torch_first", "aps_first"])
def test_import_sets_the_hardware_queues(order):
    """`import aps_amd` asks the HIP runtime for 8 hardware queues when the caller set nothing and
    the runtime is not up yet -- also when torch was imported first (the variable is read when the runtime
    initialises, not at `import torch`).  Measured, not assumed: six one-block spin kernels on six streams take about
    as long as one when every stream has a queue of its own, and twice as long on the default four queues."""
    import os
    import subprocess
    import sys
    prog = ("import torch, time\n" if order == "torch_first" else "") + """
import aps_amd, os
import torch, time
assert os.environ.get("GPU_MAX_HW_QUEUES") == "8", os.environ.get("GPU_MAX_HW_QUEUES")
streams = [torch.cuda.Stream() for _ in range(6)]
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
def run(ss, cyc=4000000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in ss:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize(); return time.perf_counter() - t0
run(streams)
one = min(run(streams[:1]) for _ in range(3)); six = min(run(streams) for _ in range(3))
print("RATIO", six / one)
"""
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", prog], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    ratio = float(out.stdout.split("RATIO")[1].split()[0])
    print(f"[hardware queues] {order}: six streams / one stream = {ratio:.2f}")
    assert ratio < 1.5, f"six streams took {ratio:.2f} x one stream: they share hardware queues"


def test_pipelined_replicas_lookahead_matches_eager(device):
    """lookahead: submit() launches the front of batch k (stage A + the persistent launch) and the back of batch
    k - workers; with stage A on the workers the head stream carries the LSTM launches only.  Same bits as the eager
    step for every resident batch, on changing inputs, and nothing is left pending behind synchronize() / wait()."""
    from aps_amd import nn_ops
    from aps_amd.replicas import PipelinedReplicas
    th.manual_seed(15)
    rnn = th.nn.LSTM(128, 128, num_layers=2, batch_first=True).eval().to(device)
    pre = th.nn.Linear(96, 128).eval().to(device)
    proj = th.nn.Linear(128, 96).eval().to(device)
    xs = [th.randn(8, 20, 96, device=device) for _ in range(6)]

    def step(x):
        h = nn_ops.linear(x, pre.weight, pre.bias)
        return nn_ops.linear(nn_ops.lstm_forward(rnn, h), proj.weight, proj.bias)

    with th.no_grad():
        with pytest.raises(ValueError):
            PipelinedReplicas([lambda x=x: step(x) for x in xs[:4]], workers=3, lookahead=True)
        reps = PipelinedReplicas([lambda x=x: step(x) for x in xs], workers=3, lstm_share=2, front="worker",
                                 mid="worker", lookahead=True)
        returned = []
        for _ in range(4 * len(reps)):
            index, _ = reps.submit()
            returned.append(index)
        assert returned[:3] == [None, None, None] and returned[3:9] == [0, 1, 2, 3, 4, 5]
        reps.synchronize()
        assert not reps._pending
        reps.check_outputs(reps.eager_outputs, "after 24 lookahead submissions")
        for x in xs:
            x.mul_(0.5)
        eager2 = [step(x) for x in xs]
        for _ in range(len(reps)):
            reps.submit()
        for b in range(len(reps)):
            assert th.equal(reps.wait(b), eager2[b])
        reps.close()
