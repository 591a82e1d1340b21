for cfg in "16 16" "16 32" "16 64" "16 128" "64 200" "2 64"; do
  set -- $cfg
  echo "== chunks of $1 MiB, pool of $2 GiB"
  timeout -k 10 200 python scripts/time_placement_slab.py vmmpool $1 $2 2>&1 | grep -v amdgpu.ids | grep "every\|pool of" | head -8
done
