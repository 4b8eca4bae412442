import sys, os, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] + "/tests")
import torch.multiprocessing as mp
import test_gpu_dp as T
if __name__ == "__main__":
    out = "/tmp/dpdiag"
    mp.spawn(T._worker, args=(2, T._free_port(), out), nprocs=2, join=True)
    r0 = torch.load(out + ".0")
    grads, param = T._run(list(range(T.FRAMES)))
    for s, (g2, g1) in enumerate(zip(r0["grads"], grads)):
        g1 = g1.cpu()
        tol = 2e-4 * g1.abs() + 2e-6 * float(g1.abs().max())
        bad = (g2 - g1).abs() > tol
        print("step", s, "bad", int(bad.sum()), "of", g1.numel(), "max diff", float((g2 - g1).abs().max()), "max", float(g1.abs().max()),
              "worst ratio", float(((g2 - g1).abs() / tol).max()))
        idx = torch.nonzero(bad).flatten()[:10]
        print(idx.tolist(), g1[idx].tolist(), g2[idx].tolist())
