import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f samples/s  %.4f ms' % (d['value'], d['ms_per_step']), d.get('step_matrix'), 'fwd_off', (d.get('fwd_roofline_ert_off') or {}).get('fused_fwd_us'))
