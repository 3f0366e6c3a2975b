"""One-screen summary of a bench.py DETAIL record (`--detail`, default bench_detail.json): python tools/bench_summary.py bench_detail.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print('c3: value %.4g  ms/step %.4f  roofline frac %.4f (%s %.4f ms)  launch %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['kernel_ms'], (d['config'].get('launch') or {}).get('mode')))
for k, v in d['roofline']['per_kernel'].items():
    print('   %-34s %.4f ms  frac %.4f' % (k, v['ms'], v['frac']))
if 'forward_only' in d:
    f = d['forward_only']; print('c3f: ms/step %.4f  kernel %.4f  frac %.4f' % (f['ms_per_step'], f['roofline']['kernel_ms'], f['roofline']['frac']))
for b, v in d['roofline'].get('batch_sweep', {}).get('batches', {}).items():
    print('   B=%-6s fwd %.4f (%.3f)  bwd %.4f (%.3f)' % (b, v['fwd_ms'], v['fwd_frac'], v['bwd_ms'], v['bwd_frac']))
for k, v in d.get('other_workloads', {}).items():
    print('%-13s ms/step %.4f  %s' % (k, v['ms_per_step'], {kk: round(vv['ms'], 4) for kk, vv in v['per_kernel'].items()}))
if d.get('cpu_baseline'):
    print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], {k: round(v['value']) for k, v in d['cpu_baseline'].get('legs', {}).items()})
print('record bytes', d['roofline'].get('record_bytes_per_launch'), 'comm_ms', d.get('comm_ms'))
