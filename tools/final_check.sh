cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
OUT=gpurun_out/r05b; mkdir -p $OUT
for v in "nodp::" "p2p:1:p2p" "rccl:1:pg"; do
  tag=${v%%:*}; rest=${v#*:}; force=${rest%%:*}; coll=${rest#*:}
  SED_FORCE_DP=$force SED_DP_COLLECTIVE=$coll timeout 300 python bench.py --steps 1000 --no-cpu-baseline --no-extras > $OUT/dp1_${tag}_bench.json 2> /dev/null
  python -c "import json; d=json.loads(open('$OUT/dp1_${tag}_bench.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'])"
done
