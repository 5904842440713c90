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
