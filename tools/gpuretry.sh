#!/bin/bash
# usage: gpuretry.sh TIMEOUT 'command'
for i in $(seq 1 15); do
  out=$(/usr/local/graft/bin/gpurun --timeout $1 -- "$2" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 60; continue; fi
  echo "$out"; exit 0
done
echo "gave up"; exit 3
