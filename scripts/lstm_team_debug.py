import os, sys
sys.path.insert(0, "/root/repo")
import torch
from aps_amd import nn_ops
torch.manual_seed(0)
N,T,D,H = 16,249,512,512
with torch.no_grad():
    rnn = torch.nn.LSTM(D, H, 1, batch_first=True).eval().cuda()
    x = torch.randn(N, T, D, device="cuda")
    for dbg in ("32", "40", "48"):
        os.environ["APS_LSTM_DEBUG"] = dbg
        st = nn_ops._lstm_status(x.device)
        st.ws[1:4] = 0
        for _ in range(3): out = nn_ops.lstm_forward(rnn, x)
        torch.cuda.synchronize()
        e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out = nn_ops.lstm_forward(rnn, x)
        e1.record(); torch.cuda.synchronize()
        w = st.ws[:4].tolist()
        print(f"debug={dbg}: {e0.elapsed_time(e1)/10*1e3/T:.2f} us/step; decisions word {w[1]:#x} xcc word {w[2]:#x} timeouts {w[0]}")
