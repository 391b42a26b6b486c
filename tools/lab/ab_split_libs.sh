#!/bin/bash
# Same-box A/B of library variants (artdeco_amd/lib/libartdeco_hip.<variant>.so) on tools/lab/ab_split.py, interleaved twice.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for so in $ROOT/artdeco_amd/lib/libartdeco_hip*.so; do
    echo "== $(basename $so) rep $rep"
    ARTDECO_HIP_LIB=$so python $ROOT/tools/lab/ab_split.py "$@" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if '{' not in line: continue
    cfg, js = line.split(' ', 1)
    d = json.loads(js)
    print(cfg, d['tiles'], ' | '.join('%s f %.4f b %.4f s %.3f' % (k[3]+k[8], min(x['raster_fwd'] for x in v), min(x['raster_bwd'] for x in v), min(x['step_ms'] for x in v)) for k, v in d.items() if k.startswith('fwd') and k != 'fwd_bit_identical'))
"
  done
done
