#!/bin/bash
# tools/ab_lib.sh libA.so libB.so [rounds]: alternate two builds of libdsir.so (DSIR_LIB) on one box
a=$1; b=$2; n=${3:-2}
for i in $(seq 1 $n); do for l in $a $b; do DSIR_LIB=$PWD/$l bash tools/ab_env.sh DSIR_X 1 | sed "s|DSIR_X=1|$l|"; done; done
