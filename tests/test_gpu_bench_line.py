"""The default `python bench.py` line at N = 1 (what the driver records as BENCH_rNN.json), at a reduced size: one JSON line with the
contract's fields, the roofline and extra-line objects, and the compact lines of the other configurations (`other_configs`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.timeout(900)
def test_default_line_carries_the_other_configurations():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gaussians", "6000", "--width", "160", "--height", "96",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                            # ONE JSON line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["dtype"] == "f32" and d["config"]["workload"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert [("error" in x) for x in d["extra_lines"]] == [False, False]
    oc = d["other_configs"]
    assert [x["equivalent_flags"] for x in oc] == ["--channels 32", "--render-iter", "--render-iter --attr-channels 4", "--per-frame", "--per-frame-fused",
                                                    "--ref-flow", "--gaussians 1000000 --width 1280 --height 720", "(knn_full)"]
    for x in oc:
        assert "error" not in x and x["value"] > 0, x
    assert oc[5]["forward_only"] > oc[5]["value"] and oc[6]["tile_pairs_M"] > 1000000
    # the real-workload figures in a compact `summary`, the LAST key of the line (the driver keeps the last
    # 2000 characters), at most 600 characters
    assert list(d)[-1] == "summary"
    sm = d["summary"]
    assert list(sm)[:12] == ["headline_fps", "training_frame_fps", "train_step_ms", "c4_fps", "c5_fps", "render_iter_fps",
                             "render_iter_attr4_fps", "per_frame_fps", "per_frame_fused_fps", "ref_flow_fwd_fps", "knn_full_ms", "fwd_only_fps"]
    assert all(isinstance(sm[k], (int, float)) and sm[k] > 0 for k in list(sm)[:12]), sm
    assert len(json.dumps(sm)) <= 600 and lines[0].rstrip().endswith(json.dumps(sm) + "}")
    assert sm["headline_fps"] == d["value"] and sm["train_step_ms"] == d["extra_lines"][1]["train_step_ms"]


@pytest.mark.timeout(600)
def test_force_process_group_runs_the_step_on_rccl_with_one_rank():
    """`bench.py --gpus 1 --force-process-group` initialises the RCCL process group (world size 1) so that
    `ranks_seen` comes from a real all-reduce and the step's collective runs; `--zero1` adds reduce-scatter / all-gather"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for extra in ([], ["--zero1"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gaussians", "6000", "--width", "160", "--height", "96",
                            "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--no-extra-lines",
                            "--force-process-group"] + extra, env=env, capture_output=True, text=True, timeout=500)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["value"] > 0
        assert ("zero1" in d["config"]["parallelism"]) == bool(extra)
