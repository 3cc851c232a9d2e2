"""bench.py against an alternative build of the library: SPIRAL_LIB=tools/variants/libspiral_X.so python tools/variant_bench.py [bench flags]
(an older build may lack entry points added since: their bindings are dropped here, and the bench legs that need them -- e.g. the
query lanes -- must be switched off on the command line, --lanes 1)"""
import ctypes, os, sys
import torch  # first: it ships its own HIP runtime, which the library must bind to
torch.cuda.is_available()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spiral_amd._lib as L
if os.environ.get("SPIRAL_LIB"):
    L.LIB_PATH = os.environ["SPIRAL_LIB"]
    have = ctypes.CDLL(L.LIB_PATH)
    for name in [n for n in L.PROTOTYPES if not hasattr(have, n)]:
        print(f"variant_bench: {os.path.basename(L.LIB_PATH)} does not export {name}", file=sys.stderr)
        del L.PROTOTYPES[name]
import bench
bench.main()
