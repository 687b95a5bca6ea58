#!/bin/bash
# A/B on one box over several library variants: tools/ab3_exp.sh "v1 v2 ..." — lone frame / 4 in flight (tools/pipeline_cost.py) and the bench line, "" = the product build
for lib in "" $1 "" $1; do
  echo "== lib=${lib:-current}"
  RT_LIB_VARIANT=$lib N_LIST=1 P_LIST=1,4 N_CTX=4 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-330
  RT_LIB_VARIANT=$lib python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   bench ms/step %.4f' % d['ms_per_step'])"
done
