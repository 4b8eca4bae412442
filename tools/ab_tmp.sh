cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_knn.py -x -q < /dev/null 2>&1 | tail -12
PYTHONPATH=. timeout 100 python - <<PY
import torch, time
from splatter_a_video_amd.knn import knn_points
from splatter_a_video_amd.synth import make_scene
from splatter_a_video_amd import _lib as L
sc = make_scene(300000, 854, 480, C=3, seed=1)
pts = torch.tensor(sc.positions(0), device="cuda")
for it in range(3):
    if it == 1:
        L.profile_reset(); L.profile_enable(True)
    torch.cuda.synchronize(); t=time.time()
    r = knn_points(pts[None], pts[None], None, None, K=6)
    torch.cuda.synchronize(); print("wall ms", (time.time()-t)*1e3)
L.profile_enable(False)
for k in ("knn_bbox","knn_plan","knn_count","knn_scatter","knn_search"):
    print(k, L.profile_read(k))
u = torch.rand(300000, 3, device="cuda")
torch.cuda.synchronize(); t=time.time(); r = knn_points(u[None], u[None], None, None, K=6); torch.cuda.synchronize(); print("uniform cube wall ms", (time.time()-t)*1e3)
PY
