"""A/B helper: run bench.py against another build of the library (SAM6D_AB_LIB=path)."""
import os, sys, runpy
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(root, "openvino-sam-6d_amd"))
from sam6d_hip import _lib
if os.environ.get("SAM6D_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SAM6D_AB_LIB"])
sys.argv = ["bench.py"] + sys.argv[1:]
os.chdir(root)
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
