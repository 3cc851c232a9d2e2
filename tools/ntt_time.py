"""ns per limb-pair transform of the batched to_ntt / from_ntt launches (spiral_gpu_time_ntt) at several batch sizes;
SPIRAL_LIB=<path> times an alternative build of the library.  tools/ntt_time.py [batch ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spiral_amd as sa
for n in [int(x) for x in sys.argv[1:]] or [256, 1024, 4096, 16384, 65536]:
    f, i = sa.time_ntt(n, 20)
    print(f"{os.environ.get('SPIRAL_LIB', 'product'):40s} batch {n:6d}: forward {f * 1e6 / n:6.2f} ns ({f * 1e3:7.1f} us)  inverse {i * 1e6 / n:6.2f} ns ({i * 1e3:7.1f} us) per launch")
