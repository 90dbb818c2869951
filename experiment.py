#!/usr/bin/env python
"""`python experiment.py --config dafnet_config_chaos --split 0 --l_mix 1` -- same entry point as the reference."""
from multimodal_segmentation_amd.experiment import Experiment

if __name__ == '__main__':
    Experiment().run()
