#!/bin/bash
# A/B of alternative libraries on family_sweep.rate cases (development aid):
#   tools/ab_family.sh "lib_a lib_b ..." "kind:proposal:dim ..."     (lib names under rwm-pt-pytorch_amd/, "lib" = shipped)
for spec in $2; do
  IFS=: read kind prop dim <<< "$spec"
  for l in $1; do
    PTRWM_LIB=$PWD/rwm-pt-pytorch_amd/$l/libptrwm_hip.so python3 -c "
import sys; sys.path.insert(0, 'tools')
import family_sweep as F
print('kind $kind $prop dim $dim  $l  %.4g' % F.rate($kind, '$prop', $dim))" 2>/dev/null
  done
done
