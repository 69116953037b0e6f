"""gfx950: a vector instruction that reads a scalar register pair written by a vector instruction needs two wait states
in between (and a DPP read of a VGPR written by a vector instruction too).  The compiler guarantees that for its own
instructions, not across inline asm.  This scans a device assembly listing (hipcc -S --cuda-device-only) block by block
and reports the smallest distance it finds per reading opcode -- `python scripts/check_sgpr_hazards.py file.s`."""
import re, sys

def operands(t, op):
    return [a.strip() for a in t[len(op):].split(',')]

def scan(path):
    pos = 0
    lastw, lastv = {}, {}
    mind, viol = {}, 0
    for raw in open(path):
        t = raw.strip()
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        if t.endswith(':'):
            lastw, lastv = {}, {}
            continue
        op = t.split()[0]
        if op == 's_nop':
            pos += int(t.split()[1]) + 1
            continue
        if op.startswith('v_'):
            ops = operands(t, op)
            reads, writes = [], []
            if op in ('v_subb_co_u32_e64', 'v_addc_co_u32_e64'):
                writes, reads = [ops[1]], [ops[-1].split()[0]]
            elif op == 'v_cndmask_b32_e64':
                reads = [ops[-1].split()[0]]
            elif op.startswith('v_cmp') and op.endswith('_e64'):
                writes = [ops[0]]
            elif op.startswith('v_cmp'):
                writes = ['vcc']
            elif op.endswith('_e32') and any(k in op for k in ('addc', 'subb', 'cndmask')):
                reads = ['vcc']
                if 'cndmask' not in op:
                    writes = ['vcc']
            elif 'add_co' in op or 'sub_co' in op:
                writes = [ops[1]]
            for r in reads:
                if r in lastw:
                    d = pos - lastw[r] - 1
                    mind[op] = min(mind.get(op, 99), d)
                    viol += d < 2
            if '_dpp' in op:  # DPP source = second operand
                src = ops[1].split()[0]
                if src in lastv:
                    d = pos - lastv[src] - 1
                    mind[op + ' (vgpr)'] = min(mind.get(op + ' (vgpr)', 99), d)
                    viol += d < 2
            for w in writes:
                lastw[w] = pos
            if ops and re.fullmatch(r'v\d+', ops[0]):
                lastv[ops[0]] = pos
        pos += 1
    return viol, mind

if __name__ == "__main__":
    bad = 0
    for f in sys.argv[1:]:
        v, m = scan(f)
        bad += v
        print(f, "violations", v, m)
    sys.exit(1 if bad else 0)
