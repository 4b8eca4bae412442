"""Opt-in import-path shim for ``simple_knn`` (reference pin: requirements.txt:83), whose only use is
``from simple_knn._C import distCUDA2`` (src/pointrix/utils/gaussian_points/gaussian_utils.py:5,70).  NOT simple_knn: see
shims/README.md."""
