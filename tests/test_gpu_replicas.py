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
