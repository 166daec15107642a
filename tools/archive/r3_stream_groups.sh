for cfg in ":" "JXL_AUX_STREAMS=0:--stream-groups 4" "JXL_AUX_STREAMS=1:--stream-groups 4" "JXL_AUX_STREAMS=1:--stream-groups 2" "JXL_AUX_STREAMS=0:--stream-groups 8" ":"; do
  e=${cfg%%:*}; a=${cfg#*:}
  env $e python bench.py --no-cpu-baseline --no-end-to-end --no-gather $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$e $a]', d['value'], d['ms_per_step'])"
done
