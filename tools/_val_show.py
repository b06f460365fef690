import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], '%.1f M reads/s  %.4f ms  path %s' % (d['value']/1e6, d['ms_per_step'], d['config'].get('kernel_path')))
