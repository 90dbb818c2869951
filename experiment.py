#!/usr/bin/env python
"""`python experiment.py --config dafnet_config_chaos --split 0 --l_mix 1` -- same entry point as the reference
(experiment.py:100-124).  Run from this directory, the reference's top-level module names (`models.dafnet`,
`model_executors.dafnet_executor`, `configuration.<name>`, `model_components.*`, ...) resolve to the MI355X package
(multimodal_segmentation_amd/compat.py)."""
import multimodal_segmentation_amd.compat as _compat

_compat.install()
from multimodal_segmentation_amd.experiment import Experiment  # noqa: E402

if __name__ == '__main__':
    Experiment().run()
