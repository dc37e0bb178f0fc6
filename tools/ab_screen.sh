mkdir -p gpurun_out
for v in ${VARIANTS:-"" norank nolds nostage all3}; do
  if [ -z "$v" ]; then unset DSIR_LIB; else export DSIR_LIB=$PWD/deepsir_amd/libdsir_$v.so; fi
  echo "== variant ${v:-base}"
  bash tools/kstat.sh v$v screen_kernel -- python3 tools/microbench.py match_screened --clouds 64 --reps 20 | grep screen_kernel
done
