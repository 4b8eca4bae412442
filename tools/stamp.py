"""Provenance stamp of a measurement record: the commit the tree was at when the GPU call was made (written HERE, before
gpurun -- the GPU box has no .git) and the build id of the library that actually ran (read on the box from the .so).
    python tools/stamp.py write      # in the build container, before gpurun: tools/.stamp.json (git-ignored, travels)
    from stamp import stamp          # on the box, in the tools that write profiles/*.json
"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "tools", ".stamp.json")


def build_id() -> str:
    lib = ctypes.CDLL(os.environ.get("SPLAT_LIB_PATH") or os.path.join(ROOT, "splatter_a_video_amd", "libsplat_hip.so"))
    lib.splat_build_id.restype = ctypes.c_char_p
    return lib.splat_build_id().decode()


def stamp() -> dict:
    s = {"git_head": None, "git_dirty": None}
    if os.path.exists(PATH):
        s.update(json.load(open(PATH)))
    s["build_id"] = build_id()
    return s


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "write":
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
    dirty = subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "splatter_a_video_amd", "include", "bench.py", "tools"],
                                    text=True)
    json.dump({"git_head": head, "git_dirty": sorted(l.split(None, 1)[1] for l in dirty.splitlines() if l.strip()) or False, "build_id_at_push": build_id()},
              open(PATH, "w"), indent=1)
    print(open(PATH).read())
