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
    assert [x["equivalent_flags"] for x in oc] == ["--channels 32", "--render-iter", "--render-iter --attr-channels 4", "--per-frame", "--ref-flow",
                                                    "--gaussians 1000000 --width 1280 --height 720"]
    for x in oc:
        assert "error" not in x and x["value"] > 0, x
    assert oc[4]["forward_only"] > oc[4]["value"] and oc[5]["tile_pairs_M"] > 1000000
