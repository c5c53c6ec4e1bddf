"""Reads bench.py's JSON line from stdin and prints the headline (and the step matrix / per-kernel times): a helper of
tools/ab.sh / tools/ab_lib.sh on the GPU box."""
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f samples/s  %.4f ms' % (d['value'], d['ms_per_step']), d.get('step_matrix'), 'fwd_off', (d.get('fwd_roofline_ert_off') or {}).get('fused_fwd_us'))
