"""per-parameter gradient error of DCCRN's train()-mode step against the reference fixtures
(tests/golden/<tag>_train_<mode>.npz), worst first.   python scripts/dccrn_grad_errors.py [case]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_dccrn import small_net  # noqa: E402
from tests.test_oracle_encoder import DCCRN_TRAIN_CASES, dccrn_train_reference  # noqa: E402

dev = torch.device("cuda:0")
cases = [int(a) for a in sys.argv[1:]] or range(len(DCCRN_TRAIN_CASES))
for c in cases:
    tag, kw, mode = DCCRN_TRAIN_CASES[c]
    sd, mix, ref = dccrn_train_reference(tag, kw, mode)
    net = small_net(**dict(kw))
    net.training_mode = mode
    net.load_state_dict(sd, strict=False)
    net = net.train().to(dev)
    out = net(mix.to(dev))
    loss = sum((o * ref[f"probe{s}"].to(dev)).sum() for s, o in enumerate(out))
    loss.backward()
    rows = []
    for k, p in net.named_parameters():
        if "grad." + k not in ref or k.endswith(("block.0.real.bias", "block.0.imag.bias", "block.0.bias")):
            continue
        want = ref["grad." + k].double()
        err = ((p.grad.cpu().double() - want).abs().max() / want.abs().max()).item()
        rows.append((err, k))
    oe = max(((o.detach().cpu().double() - ref[f"out{s}"].double()).abs().max() /
              ref[f"out{s}"].abs().max()).item() for s, o in enumerate(out))
    print(f"== {tag} {mode}: output error {oe:.2e}")
    for err, k in sorted(rows, reverse=True)[:10]:
        print(f"   {k:55s} {err:.2e}")
