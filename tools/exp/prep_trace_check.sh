cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_rig.py tests/test_gpu_emit_maps.py -q -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for C in 3 5; do
  O=$GRAFT_REPO_ROOT/gpurun_out/prep_trace_$C; rm -rf $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 200 --config $C --only emit > $O.log 2>&1
  f=$(find $O -name '*kernel_stats.csv' | head -1); grep -i "chain_prep\|emit_multi" $f | cut -c1-140
  find $O -name '*.csv' -size +1M -delete
done
