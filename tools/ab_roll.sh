#!/bin/bash
# Dev helper: same-box A/B of the rolling CReFF kernel -- every library given (default: the shipped one) runs tools/check_roll.py --only-big,
# interleaved, three rounds; prints ms per frame of the 512x1024 x 11-frame launch per library and round (boxes differ by ~7 %: compare within one call).
#   bash tools/ab_roll.sh scratch/rr_libs/lib_base.so ar-seg_amd/lib/libarseg_hip.so
for round in 1 2 3; do
  for lib in "$@"; do
    ARSEG_HIP_LIB=$lib python tools/check_roll.py --only-big --iters 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', 'round $round', 'roll_ms_per_frame %.4f' % d['roll_ms_per_frame'], 'max_abs_p %.2e' % d['max_abs_p'])
"
  done
done
