cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r05c
NT=visgeom_amd/lib/variants/libvisgeom_amd_nt.so
{
for rep in 1 2; do
  timeout 300 python tools/exp/emit_sweep_probe.py
  AB_LIB=$NT timeout 300 python tools/exp/emit_sweep_probe.py
done
INLINE_MAX=1 timeout 300 python tools/exp/emit_sweep_probe.py
INLINE_MAX=100000000000 timeout 300 python tools/exp/emit_sweep_probe.py
AB_LIB=$NT INLINE_MAX=1 timeout 300 python tools/exp/emit_sweep_probe.py
AB_LIB=$NT INLINE_MAX=100000000000 timeout 300 python tools/exp/emit_sweep_probe.py
} > gpurun_out/${T}_emit_sweep_ab.txt 2>&1
cat gpurun_out/${T}_emit_sweep_ab.txt | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_c3 -o t -- python $R/tools/bench_configs.py 200 --config 3 --only emit > $R/gpurun_out/${T}_prof_c3.log 2>&1
echo "trace rc=$?"; find $R/gpurun_out/${T}_prof_c3 -name '*kernel_stats.csv' | head -1 | xargs cat | head -8
find $R/gpurun_out/${T}_prof_c3 -name '*kernel_trace.csv' -delete
