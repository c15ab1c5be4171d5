for t in '{}' '{"steps_per_visit": 200}' '{"brick_cells": 36864, "threads": 1024, "steps_per_visit": 200}' '{"brick_cells": 8704, "steps_per_visit": 200}' '{"steps_per_visit": 200, "park_below": 1}' '{"steps_per_visit": 200, "chunk": 4096}'; do
  echo "TUNE $t"
  timeout -k 10 200 python tools/exp_sca.py --exec 1 --cl-global 49526352 --ps-global 4194304 --tune "$t" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-26s %8.1f ms  %.3e packets/s  %.3e steps/s  passes %d' % (d['launch'], d['kernel_ms'], d['packets_per_s'], d.get('steps_per_s', 0), d.get('passes', 0)))"
done
