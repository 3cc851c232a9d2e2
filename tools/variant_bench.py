"""bench.py against an alternative build of the library: SPIRAL_LIB=tools/variants/libspiral_X.so python tools/variant_bench.py [bench flags]
(spiral_amd/_lib.py honours SPIRAL_LIB itself, so this is plain bench.py -- kept for the command lines recorded under profiles/; the
ranks a multi-GPU run launches inherit the variable.  An older build may lack entry points added since: their bindings are dropped with a
message and the bench legs that need them, e.g. the query lanes, must be switched off on the command line, --lanes 1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.main()
