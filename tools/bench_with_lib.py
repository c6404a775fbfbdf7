"""A/B runs: bench.py against an alternative build of the library (DVAE_EXP_LIB=path)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _dvae_pkg import dvae_amd
if os.environ.get("DVAE_EXP_LIB"):
    dvae_amd._lib.LIB_PATH = os.path.abspath(os.environ["DVAE_EXP_LIB"])
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
