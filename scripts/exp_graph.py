"""One step of the full chain captured in a HIP graph (torch.cuda.CUDAGraph around m17gpu_rx_blocks) against plain
launches: same outputs, step time by events.   python scripts/exp_graph.py [C] [nblk] [afc]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 12
afc = int(sys.argv[3]) if len(sys.argv) > 3 else 0
T = 8
gen = m.Receiver(C, nblk)
big = gen.gen_batch(nblk * T)["iq"]
slabs = big.view(C, T, nblk, 1920, 2).permute(1, 0, 2, 3, 4).contiguous()
gen.close()

def run(graph):
    rx = m.Receiver(C, nblk)
    rx.set_option("afc", afc)
    out = rx.alloc_outputs(nblk)
    stage = torch.empty_like(slabs[0])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        stage.copy_(slabs[0]); rx.rx_blocks(stage, 1, out)          # warm-up outside the capture
        torch.cuda.synchronize()
        g = None
        if graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                rx.rx_blocks(stage, 1, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        res = []
        tt = 0.0
        for k in range(1, T):
            stage.copy_(slabs[k])
            e0.record()
            if g is not None: g.replay()
            else: rx.rx_blocks(stage, 1, out)
            e1.record()
            torch.cuda.synchronize()
            tt += e0.elapsed_time(e1)
            res.append((out["recs"].clone(), out["counts"].clone()))
    rx.close()
    return tt / (T - 1), res

t_plain, r_plain = run(False)
t_graph, r_graph = run(True)
same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(r_plain, r_graph))
print(f"C={C} nblk={nblk} afc={afc}: plain launches {t_plain:.4f} ms per step, graph replay {t_graph:.4f} ms; outputs identical: {same}")
