"""Two independent forwards (two model copies, two stream pairs) in flight at once against one after the other:
does the graph have idle capacity that a second batch can use?  python tools/ubench/two_in_flight.py"""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from decnet_amd.model import get_model

dev = torch.device("cuda:0")
B = 8
torch.manual_seed(17)
mk = lambda: get_model(name="sparsedensenetrefinementmask", max_disp=bench.MAX_DISP, base_channels=8, cost_func="cor",
                       grad_method="detach", num_stage=4, down_scale=3, step=[-1., 1., 1., 1.],
                       samp_num=[-1., 12., 10., 6.], sample_spa_size_list=[-1, 3, 5, 7], down_func_name="bicubic",
                       weights=[1., 1., 1., 1.], if_overmask=False, skip_stage_id=4, use_detail=True, thold=0.5).to(dev).eval()
m1 = mk()
m2 = copy.deepcopy(m1)
g = torch.Generator(device=dev).manual_seed(17)
ins = [(torch.randn(B, 3, bench.PAD_H, bench.PAD_W, device=dev, generator=g),
        torch.randn(B, 3, bench.PAD_H, bench.PAD_W, device=dev, generator=g)) for _ in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
iters = 10
with torch.no_grad():
    for _ in range(3):
        m1(*ins[0]); m2(*ins[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        m1(*ins[0]); m2(*ins[1])
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) / iters
    print("one after the other (eager, one stream): %.3f ms per two batches" % (1e3 * seq))
    # graphs: one per model, captured on its own stream
    graphs = []
    for m, x, s in ((m1, ins[0], s1), (m2, ins[1], s2)):
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(*x)
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            m(*x)
        graphs.append(gr)
    torch.cuda.synchronize()
    for _ in range(2):
        graphs[0].replay(); graphs[1].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        graphs[0].replay(); graphs[1].replay()
    torch.cuda.synchronize()
    print("graph replays from one stream: %.3f ms per two batches" % (1e3 * (time.perf_counter() - t0) / iters))
    t0 = time.perf_counter()
    for _ in range(iters):
        with torch.cuda.stream(s1):
            graphs[0].replay()
        with torch.cuda.stream(s2):
            graphs[1].replay()
    torch.cuda.synchronize()
    print("graph replays on two streams: %.3f ms per two batches" % (1e3 * (time.perf_counter() - t0) / iters))
    # eager, two streams, launches of the two forwards interleaved by two host threads
    import threading
    def run(m, x, s, n):
        with torch.no_grad(), torch.cuda.stream(s):
            for _ in range(n):
                m(*x)
    for n in (2, iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(m, x, s, n)) for m, x, s in ((m1, ins[0], s1), (m2, ins[1], s2))]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print("eager on two streams from two host threads: %.3f ms per two batches" % (1e3 * dt))
