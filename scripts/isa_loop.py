"""Static instruction mix of k_evaluate's time-step loop in /tmp/isa/fot.s (scripts/isa.sh): per basic block of the
loop the VALU / SALU / LDS / SMEM counts, so that two builds can be compared block by block."""
import re
import sys

t = open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/isa/fot.s').read()
kern = sys.argv[2] if len(sys.argv) > 2 else 'k_evaluateE'
m = re.search(r'\n(_ZN3fot\d+' + kern + r'[^\n]*):\s*;[^\n]*\n(.*?)\n\.Lfunc_end', t, re.S)
body = m.group(2).split('\n')
def children(i):
    n = 0
    while 'Child Loop' in body[i + 1 + n]:
        n += 1
    return n
# the time-step loop: the innermost loop header that still has five or more child loops (chunk walks, re-checks)
heads = [(int(re.search(r'Depth=(\d+)', l).group(1)), i) for i, l in enumerate(body) if 'Loop Header: Depth=' in l and children(i) >= 5]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1           # several time-step loops in one kernel: which one
start = sorted(h for h in heads if h[0] == max(heads)[0])[which][1]
while not re.match(r'\.LBB\d+_\d+:', body[start]):       # (the label sits a few comment lines above a nested header)
    start -= 1
label = body[start].split(':')[0]
end = max(i for i, l in enumerate(body) if re.search(r's_c?branch\w*\s+' + re.escape(label) + r'\b', l))
blk, rows, tot = 'head', [], {}
cnt = {}
def flush():
    if cnt:
        rows.append((blk, dict(cnt)))
for l in body[start:end + 1]:
    s = l.strip()
    if not s or s.startswith(';'):
        if s.startswith('; %bb.'):
            flush(); cnt = {}; blk = s.split(':')[0][2:]
        continue
    if re.match(r'\.LBB\d+_\d+:', s):
        flush(); cnt = {}; blk = s.split(':')[0]
        continue
    op = s.split()[0]
    kind = ('VALU' if op.startswith('v_') else 'SALU' if op.startswith('s_') and not op.startswith('s_load') and not op.startswith('s_waitcnt')
            else 'SMEM' if op.startswith('s_load') else 'LDS' if op.startswith('ds_') else 'VMEM' if op.startswith(('global_', 'flat_', 'buffer_')) else 'other')
    cnt[kind] = cnt.get(kind, 0) + 1
    if kind == 'VALU':
        sub = 'f64' if '_f64' in op and not op.startswith('v_cmp') and not op.startswith('v_cvt') else 'pk32' if op.startswith('v_pk_') else 'f32' if '_f32' in op and not op.startswith('v_cmp') else 'lane' if 'lane' in op else 'cmp' if op.startswith('v_cmp') else 'mov' if op.startswith('v_mov') else 'cnd' if op.startswith('v_cndmask') else 'int'
        cnt[sub] = cnt.get(sub, 0) + 1
flush()
keys = ['VALU', 'f64', 'f32', 'pk32', 'cmp', 'cnd', 'mov', 'lane', 'int', 'SALU', 'SMEM', 'LDS', 'VMEM']
print('%-12s' % 'block' + ''.join('%6s' % k for k in keys))
for b, c in rows:
    if c.get('VALU', 0) + c.get('SALU', 0) >= 4:
        print('%-12s' % b + ''.join('%6d' % c.get(k, 0) for k in keys))
    for k in keys:
        tot[k] = tot.get(k, 0) + c.get(k, 0)
print('%-12s' % 'loop total' + ''.join('%6d' % tot.get(k, 0) for k in keys))
