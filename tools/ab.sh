for i in 1 2 3; do
SPIRAL_LIB=$GRAFT_REPO_ROOT/spiral_amd/libspiral_gpu_A.so python tools/stage_ab.py ""
python tools/stage_ab.py ""
done
for i in 1 2; do
SPIRAL_LIB=$GRAFT_REPO_ROOT/spiral_amd/libspiral_gpu_A.so python tools/batch_query.py 4
python tools/batch_query.py 4
done
