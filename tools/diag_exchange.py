"""diagnostic (GPU box, two gloo ranks on one device): the position-exchange training step against the all-reduce step, per parameter"""
import os, sys, torch, torch.multiprocessing as mp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_dp as T
if __name__ == "__main__":
    for steps in (1, 3):
        px, pd = f"/tmp/x{steps}", f"/tmp/d{steps}"
        mp.spawn(T._train_worker, args=(2, T._free_port(), px, True, steps, True, True), nprocs=2, join=True)
        mp.spawn(T._train_worker, args=(2, T._free_port(), pd, False, steps, True, False), nprocs=2, join=True)
        x0, x1, d0 = torch.load(px + ".0"), torch.load(px + ".1"), torch.load(pd + ".0")
        print("steps", steps, "ranks equal", torch.equal(x0["param"], x1["param"]))
        for k, (a, b) in x0["slices"].items():
            dd = (x0["param"][a:b] - d0["param"][a:b]).abs()
            print(f"  {k:16s} max diff {float(dd.max()):.3e}  frac>1e-6 {float((dd > 1e-6).float().mean()):.3f}")
