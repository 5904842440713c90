import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda:0")
cpu, w = bench.build_workload(dev, 0)
w["enh"].nan_policy = "manual"
w["enh"]._nan_guard.pointer(dev)
for two in (False, True):
    st = bench.Stages(w, two_streams=two)
    with torch.no_grad():
        for _ in range(5): st.step()
        torch.cuda.synchronize()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            st.step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st.step()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300): g.replay()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print(f"graph two_streams={two}: {el/300*1e6:.1f} us/step -> {32*300/el:.0f} utt/s")
