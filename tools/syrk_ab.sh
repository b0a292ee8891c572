# A/B of diagnostic builds of the dense-slot SYRK (timing only): tools/syrk_ab.sh name1 name2 ...  (tools/libs/lib_<name>.so)
# prints the kernel times (rocprofv3 --kernel-trace --stats) of rank 0 at 1 rank
for n in product "$@"; do
  if [ $n = product ]; then unset DROID_HIP_LIB; else export DROID_HIP_LIB=$GRAFT_REPO_ROOT/tools/libs/lib_$n.so; fi
  bash tools/prof_scale.sh 1 | grep "syrk" | sed "s/^/$n: /" | cut -c1-140
done
