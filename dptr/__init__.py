"""Drop-in shim: the reference's renderers and trainer do ``import dptr.gs as gs``
(reference: src/pointrix/renderer/dptr_ortho_enhanced.py:3, src/trainer_fragGS.py:29).
Putting this repository on PYTHONPATH makes that import resolve to the MI355X-native operators."""
from . import gs  # noqa: F401
