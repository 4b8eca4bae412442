"""scratch: loss curve / densification statistics of the composed training step on the test clip (tuning of the convergence test)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from splatter_a_video_amd import train_step as TS
from test_gpu_train_step import _clip, _perturbed, _t

def run(tag, lr_scale, thr, extent, iters=400, interval=60, start=50, stop=200):
    N, W, H, T, F = 4000, 128, 96, 20, 5
    sc, clock, truth = _clip(N, W, H, T, seed=5)
    extr = _t(sc.extr)
    rng = np.random.default_rng(0)
    cfg = TS.DensifyConfig(interval=interval, start_iter=start, grad_threshold=thr, percent_dense=1e-3, cameras_extent=extent, min_opacity=0.02, seed=123, stop_iter=stop)
    lr = {k: v * lr_scale for k, v in dict(TS.REFERENCE_LR, pos_cubic_node=2e-3, shs=2e-2, attrs=2e-2, scaling=1e-2, rotation=5e-3).items()}
    st = TS.TrainingStep(_perturbed(truth, 1), clock, W, H, F, extr, lr=lr, densify=cfg, K=8, arap_samples=256)
    cache = {}
    losses, counts = [], [st.N]
    for it in range(iters):
        t1 = [int(t) for t in rng.choice(T, F, replace=False)]
        t2 = [int(rng.choice([t for t in range(T) if t != a])) for a in t1]
        key = (tuple(t1), tuple(t2))
        if key not in cache:
            cache[key] = TS.render_ground_truth(truth, clock, W, H, extr, t1, t2)
        st.step(t1, t2, cache[key])
        losses.append(st.loss())
        c = st.cfg
        if c.start_iter < st.iteration and st.iteration % c.interval == 0:
            g = (st.dstate.pos_gradient_accum / st.dstate.denom.clamp(min=1)).flatten()
            q = torch.quantile(g, torch.tensor([0.5, 0.8, 0.9, 0.95, 0.99], device=g.device)).tolist()
            print(tag, "it", st.iteration, "grad quantiles 50/80/90/95/99:", ["%.2e" % x for x in q], "parts", {k: float(v) for k, v in st.last.items()})
        if st.maybe_densify():
            counts.append(st.N)
            print(tag, "densify ->", st.last_change)
    print(tag, "loss", ["%.4f" % np.mean(losses[i:i + 10]) for i in range(0, iters, 40)], "counts", counts)

run("thr2e-4", 1.0, 2e-4, 60.0)
run("thr2e-4_lr.5", 0.5, 2e-4, 60.0)
run("thr5e-4", 1.0, 5e-4, 60.0, iters=300, interval=50, start=40, stop=160)
