"""Markdown tables of a bench.py detail record (README / DESIGN section 5): python tools/summarize_detail.py bench_detail.json"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d['roofline']
print(f"headline: {d['ms_per_step']:.4f} ms/step = {d['value'] / 1e9:.3f} G rollout-steps/s; dominant {r['kernel']} {r['kernel_ms']:.4f} ms: frac {r['frac']:.3f} (SURVEY 8d {r['bytes_per_rollout_step']} B), "
      f"frac_model {r.get('frac_model') or 0:.3f} ({r.get('model_bytes_per_rollout_step')} B), frac_traffic {r.get('frac_traffic')}, traffic {r.get('traffic')}; {r.get('regime')}")
for k, v in r['per_kernel'].items():
    print(f"   {k:34s} {v['ms']:.4f} ms  frac {v['frac']:.3f}  frac_model {v.get('frac_model')}  traffic {v.get('traffic')} frac_traffic {v.get('frac_traffic')}")
sw = r.get('batch_sweep')
if sw:
    print('\n| rollouts | forward kernel (all six outputs) | `frac` 304 B / model / PMC | backward kernel (fit step) | `frac` 640 B / model / PMC | fit step | bound (backward) |')
    print('|---|---|---|---|---|---|---|')
    f = lambda x: '–' if x is None else f'{100 * x:.0f} %'
    for b, v in sw['batches'].items():
        print(f"| {int(b):,} | {v['fwd_ms']:.3f} ms | {f(v['fwd_frac'])} / {f(v['fwd_frac_model'])} / {f(v.get('fwd_frac_traffic'))} | {v['bwd_ms']:.3f} ms | {f(v['bwd_frac'])} / {f(v['bwd_frac_model'])} / {f(v.get('bwd_frac_traffic'))} | "
              f"{v.get('step_ms', 0):.3f} ms | {(v.get('bwd_bound') or '')[:60]} |".replace(',', ' '))
    print('first_B_at_40pct', sw['first_B_at_40pct'])
    for b, v in sw['batches'].items():
        print('   ', b, {k: round(x, 4) for k, x in v['kernels']['step'].items()}, (v['kernels'].get('rollout_bwd_kernel') or '')[:100])
ps = r.get('points_sweep')
if ps:
    print('\n| N x B | states-only forward | `frac` (80 + 32 N) | backward | `frac` (160 + 120 N) / model | backward kernel |')
    print('|---|---|---|---|---|---|')
    for k, v in ps['rows'].items():
        print(f"| {k} | {v['fwd_ms']:.3f} ms | {100 * v['fwd_frac']:.0f} % | {v['bwd_ms']:.3f} ms | {100 * v['bwd_frac']:.0f} % / {100 * (v.get('bwd_frac_model') or 0):.0f} % | {(v['kernels']['rollout_bwd_kernel'] or '')[4:60]} |")
print()
for k, v in d.get('other_workloads', {}).items():
    pk = {kk: round(vv['ms'], 4) for kk, vv in v.get('per_kernel', {}).items() if isinstance(vv, dict) and 'ms' in vv}
    print(f"{k:13s} {v['ms_per_step']:.4f} ms/step  {pk}  {(v.get('launch') or {}).get('mode', '')}")
a = d.get('other_workloads', {}).get('c3_api')
if a:
    print('c3_api cached', {k: round(v, 1) for k, v in a['cached']['host_us_per_call'].items()}, 'device', round(a['cached']['device_ms_per_step'], 4),
          '| launch by launch', round(a['launch_by_launch']['ms_per_step'], 4), {k: round(v, 1) for k, v in a['launch_by_launch']['host_us_per_call'].items()})
if d.get('forward_only'):
    print('forward_only', round(d['forward_only']['ms_per_step'], 4), d['forward_only']['roofline'])
if d.get('cpu_baseline'):
    c = d['cpu_baseline']; print('cpu_baseline', round(c['value']), c['cores'], {k: round(v['value']) for k, v in c.get('legs', {}).items()})
