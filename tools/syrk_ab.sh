# A/B of diagnostic builds of the dense-slot SYRK (timing only): tools/syrk_ab.sh name1 name2 ...  (tools/libs/lib_<name>.so)
for n in product "$@"; do
  if [ $n = product ]; then unset DROID_HIP_LIB; else export DROID_HIP_LIB=$GRAFT_REPO_ROOT/tools/libs/lib_$n.so; fi
  python tools/scale_emul.py worlds=1,8 2>/dev/null | sed "s/^/$n: /" | cut -c1-200
done
