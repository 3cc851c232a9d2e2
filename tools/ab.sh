for i in 1 2 3; do
for V in ns0 ns1 ns2; do SPIRAL_LIB=$GRAFT_REPO_ROOT/spiral_amd/libspiral_gpu_$V.so python tools/sweep_batch_time.py; done
done
