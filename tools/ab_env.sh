#!/bin/bash
# same-box A/B of an environment switch: tools/ab_env.sh VAR -- command ...   (runs: without, with, without, with)
var=$1; shift; shift
for rep in 1 2; do
  echo "== $var unset"; env -u $var "$@"
  echo "== $var=1"; env $var=1 "$@"
done
