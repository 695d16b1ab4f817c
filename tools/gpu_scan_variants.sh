R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6c3; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for rep in 1 2; do for L in lib_k2 lib_k4 lib_k6 lib_k8 lib_old; do
CF_LIB_DIR=$R/co_fusion_amd/$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$L -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20 > /dev/null 2> $O/p_$L.log
echo "$L $(python $R/tools/prof_summary.py $O/p_$L 2>&1 | grep -E 'scan_scatter' | cut -c1-40,75-130)"; rm -rf $O/p_$L
done; done | tee $O/scan_variants.txt
