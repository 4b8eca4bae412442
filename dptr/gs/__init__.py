"""``dptr.gs`` -> splatter_a_video_amd.gs (same names, argument order and defaults as
reference: src/submodules/dptr/dptr/gs/__init__.py)."""
from splatter_a_video_amd.gs import *  # noqa: F401,F403
from splatter_a_video_amd.gs import __all__  # noqa: F401
