import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "openvino-sam-6d_amd")
for _p in (ROOT, PKG, os.path.join(PKG, "pem"), os.path.join(PKG, "ism")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

GOLD = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
