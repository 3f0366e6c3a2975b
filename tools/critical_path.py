"""Two lower bounds for a loop of a gfx950 kernel from its assembly listing (hipcc -S --cuda-device-only), for ONE wave on a SIMD:

  issue bound       sum of the issue costs of the loop body's instructions (a single wave issues in order, one instruction at a time)
  recurrence bound  the longest latency-weighted dependence chain from the registers that are live around the back edge to their
                    redefinition -- what the loop-carried recurrence allows even with unlimited issue width

with the per-instruction costs measured on MI355X by tools/microbench/{pk_latency,dpp_latency}.hip
(profiles/r1g_microbench_valu_issue.txt, profiles/r2f_microbench_dpp_latency.txt), shader cycles, one wave per SIMD:

  plain VALU        issue 4.25, result usable 6.4 cycles after issue
  VALU with a DPP operand / v_mov_dpp
                    issue 6.3, result usable 12.3 cycles after issue
  transcendental (v_rcp / v_rsq / v_sqrt / v_exp / v_sin / v_cos)
                    issue 8.5 (quarter rate: two issue slots), result 16
  SALU / s_nop / branch / s_waitcnt
                    issue 4.5 (a scalar instruction takes a whole issue slot when the SIMD holds one wave), result 4.5
  LDS read          issue 4.5, result 64 (ds_read_b128: ~14 cycles of issue-equivalent when used soon after; latency ~64)
  global load       issue 4.5, result 500 (L2 hit; the kernels request a whole step ahead, so a load's consumer sits in the
                    NEXT iteration -- the chain through memory is cut at the loop boundary by construction)
  store / atomic    issue 4.5, no result

    python tools/critical_path.py file.s <kernel-name-substring> [min_loop_instrs]

Prints, per loop of at least `min_loop_instrs` instructions: the counts, both bounds per trip, and the chain itself (opcodes).
The measured time per trip sits between max(issue, recurrence) and their sum; where issue >> recurrence the loop is bound by
its instruction COUNT (no second instruction stream could be hidden under the first), where recurrence ~ issue there is slack
a co-resident wave could use."""
import re
import sys

COST = {  # kind: (issue, latency)
    'valu': (4.25, 6.4), 'dpp': (6.3, 12.3), 'trans': (8.5, 16.0), 'salu': (4.5, 4.5), 'lds': (4.5, 64.0), 'vmem': (4.5, 500.0),
    'store': (4.5, 0.0),
}
TRANS = ('v_rcp', 'v_rsq', 'v_sqrt', 'v_exp', 'v_sin', 'v_cos', 'v_log')


def kind(op, text):
    if op.startswith(('global_store', 'global_atomic', 'flat_store', 'buffer_store', 'ds_write', 'scratch_store')):
        return 'store'
    if op.startswith(('global_load', 'flat_load', 'buffer_load', 'scratch_load', 's_load', 's_buffer_load')):
        return 'vmem'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith(TRANS):
        return 'trans'
    if 'quad_perm' in text or 'row_' in text or op.endswith('_dpp'):
        return 'dpp'
    return 'valu'


def regs(tok):
    """Register names an operand token covers: v12 -> [v12]; v[4:7] -> v4..v7; s[2:3]; vcc; exec."""
    out = []
    for m in re.finditer(r'\b([vsa])\[(\d+):(\d+)\]', tok):
        out += [f'{m.group(1)}{i}' for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    for m in re.finditer(r'(?<![\w\[])([vsa])(\d+)\b', tok):
        out.append(f'{m.group(1)}{m.group(2)}')
    for name in ('vcc', 'exec', 'scc'):
        if re.search(r'\b' + name + r'\b', tok):
            out.append(name)
    return out


def parse(line):
    line = line.split(';')[0].strip()
    if not line or line.startswith('.') or line.endswith(':'):
        return None
    op, _, rest = line.partition(' ')
    ops = [t.strip() for t in re.split(r',(?![^\[]*\])', rest)] if rest.strip() else []
    k = kind(op, line)
    if k == 'store' or op.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch', 's_barrier', 's_sleep', 's_endpgm', 's_cmp', 'v_cmp')):
        dst = ['vcc'] if op.startswith('v_cmp') and ops and not ops[0].startswith('s') else (regs(ops[0]) if op.startswith('v_cmp') and ops else [])
        if op.startswith('s_cmp'):
            dst = ['scc']
        src = [r for t in (ops[1:] if op.startswith('v_cmp') else ops) for r in regs(t)]
        return op, k, dst, src
    dst = regs(ops[0]) if ops else []
    src = [r for t in ops[1:] for r in regs(t)]
    if op.startswith(('v_fmac', 'v_mac')) or 'dpp' in k and 'bound_ctrl:0' in line:      # accumulate into / keep the old destination
        src += dst
    if op.startswith(('v_cndmask', 'v_addc', 'v_subb')) and 'vcc' not in src and len(ops) < 4:
        src.append('vcc')
    return op, k, dst, src


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s+s_endpgm', text, re.S | re.M):
        if want not in m.group(1):
            continue
        lines = [l.strip() for l in m.group(2).split('\n')]
        labels = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
        print(m.group(1))
        for i, l in enumerate(lines):
            mm = re.match(r's_c?branch\w* (\.LBB\d+_\d+)', l)
            if not (mm and mm.group(1) in labels and labels[mm.group(1)] < i):
                continue
            body = [p for p in (parse(x) for x in lines[labels[mm.group(1)]:i + 1]) if p]
            if len(body) < min_n:
                continue
            counts = {}
            for _, k, _, _ in body:
                counts[k] = counts.get(k, 0) + 1
            issue = sum(COST[k][0] for _, k, _, _ in body)
            # recurrence: ready[r] = cycles (from the start of the trip) at which register r holds its new value, starting from the
            # loop-carried registers at 0; an instruction starts when its sources are ready (unlimited issue width)
            ready, via = {}, {}
            defined = set()
            carried = set()
            for op, k, dst, src in body:
                for r in src:
                    if r not in defined:
                        carried.add(r)
                defined.update(dst)
            best, best_reg = 0.0, None
            for idx, (op, k, dst, src) in enumerate(body):
                start, frm = 0.0, None
                for r in src:
                    if r in ready and ready[r] > start:
                        start, frm = ready[r], r
                if k == 'vmem':
                    done = None       # consumed in the next trip by construction (requested a step ahead): cuts the chain
                else:
                    done = start + COST[k][1]
                for r in dst:
                    if done is None:
                        ready.pop(r, None)
                    else:
                        ready[r] = done
                        via[(r, done)] = (op, frm, start)
                        if r in carried and done > best and r[0] == 'v':
                            best, best_reg = done, r
            chain = []
            r, t = best_reg, best
            while r is not None and (r, t) in via and len(chain) < 400:
                op, frm, start = via[(r, t)]
                chain.append(op)
                r, t = frm, start
            print(f'  loop of {len(body)} instructions: {counts}')
            print(f'    issue bound {issue:.0f} cycles / trip; recurrence bound {best:.0f} cycles / trip (chain of {len(chain)} instructions ending in {best_reg})')
            print('    chain: ' + ' <- '.join(chain[:60]) + (' ...' if len(chain) > 60 else ''))


if __name__ == '__main__':
    main()
