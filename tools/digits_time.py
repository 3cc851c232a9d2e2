"""ns per transform of the gadget-digit launch (spiral_gpu_time_ntt_digits) for several (source polynomials, digits) shapes;
usage: tools/digits_time.py [fwd2=0|1]   (library option fwd2: 1 = two digits per workgroup always, 0 = never); SPIRAL_LIB selects another build"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spiral_amd as sa
for a in sys.argv[1:]:
    k, v = a.split("=")
    sa.set_option(k, int(v))
tag = ",".join(sys.argv[1:]) or "defaults"
for npolys, nd in [(64, 8), (256, 8), (463, 8), (2048, 8), (16, 56), (61, 56), (2120, 56), (8192, 4)]:
    ms = min(sa.time_ntt_digits(npolys, nd, 20) for _ in range(3))
    print(f"{tag:30s} {npolys:5d} polys x {nd:2d} digits = {npolys * nd:6d} transforms: {ms * 1e3:8.1f} us  {ms * 1e6 / (npolys * nd):6.2f} ns/transform", flush=True)
