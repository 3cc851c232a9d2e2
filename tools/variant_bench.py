"""bench.py against an alternative build of the library: SPIRAL_LIB=tools/variants/libspiral_X.so python tools/variant_bench.py [bench flags]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spiral_amd._lib as L
if os.environ.get("SPIRAL_LIB"):
    L.LIB_PATH = os.environ["SPIRAL_LIB"]
import bench
bench.main()
