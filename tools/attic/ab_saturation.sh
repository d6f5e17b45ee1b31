for c in "X=1" "JT_REGION_ROT=3 JT_NO_STAGED_FINISH=1" "X=1" "JT_REGION_ROT=3 JT_NO_STAGED_FINISH=1" "X=1" "JT_REGION_ROT=3"; do
  echo "== $c"
  env $c python bench.py --steps 2 --warmup 1 --cpu-sample 0 --e2e 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['saturation']['md5']['ms_per_file'], d['saturation']['no_md5']['ms_per_file'])"
done
