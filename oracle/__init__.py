"""CPU oracle for the DAFNet / MMSDNet training step.

TEST INFRASTRUCTURE ONLY.  This package is a plain torch-CPU / numpy restatement
of the reference's algorithm (agis85/multimodal_segmentation, Keras 2.1.6 /
TF 1.4) for the hot path named in BASELINE.json.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / reported CPU baseline.  The product package
``multimodal_segmentation_amd`` never imports it.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
this path, and Keras 2.1.6 / TensorFlow 1.4 / keras-contrib 2.0.8 are neither
vendored in the reference tree nor installed here (ordinary ModuleNotFoundError,
nothing was refused), so the restatement cannot be checked against outputs of
the reference itself.  It is pinned instead by (i) known-answer tests derived
from the maths, (ii) two independent restatements of each small op (torch vs
numpy loops) and (iii) fixtures captured from the two importable reference
helpers (utils/data_utils.py::sample, utils/distributions.py) -- see
tests/golden/README.md.  Third-party defaults restated from memory are marked
with a double dagger in the docstrings.
"""
