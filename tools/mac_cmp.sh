# per-kernel durations of the expansion MAC variants (tuning): tools/mac_cmp.sh, run from the repo root on the GPU box
export TMPDIR=/tmp
O=gpurun_out/mac_cmp; mkdir -p $O
for cfg in "100000 100000" "64 100000" "64 128" "32 64"; do
  set -- $cfg
  echo "CT2_MIN=$1 CT4_MIN=$2"
  export SPIRAL_MAC_CT2_MIN=$1 SPIRAL_MAC_CT4_MIN=$2
  rm -rf $O/kt; timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/b.log 2>&1
  f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
  grep -E "expand_mac|Name" $f | cut -c1-220
done
