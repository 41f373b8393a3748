"""Build-time guard for the evaluation kernels' hand-issued scalar loads (run by csrc/Makefile on every build).

FusedSink (csrc/fot_kernels.hip) issues `s_load_dwordx16` / `s_load_dword` in one inline-asm statement and waits for them
in a later one, so that the chunk being tested and the next one overlap.  Between issue and wait the destination
registers are IN FLIGHT and the compiler does not know it: anything it does with them there -- a spill to a VGPR lane, a
copy, a reuse as a temporary, a callee that clobbers them -- reads stale data or is overwritten when the load lands
(wrong chunk data, or a wild address and a GPU fault).  This script proves, on the gfx950 ISA of the build at hand, that
no such instruction exists:

  * a forward data-flow over the control-flow graph of every k_evaluate* kernel carries the set of SGPRs in flight
    (union at joins, iterated to a fixed point, so loops and both arms of every branch are covered);
  * every instruction OUTSIDE the inline-asm statements that names an SGPR in flight, as source or destination, is
    reported -- not only the register allocator's v_writelane / v_readlane / s_mov;
  * a call (s_swappc_b64) with loads in flight is checked against every SGPR the callee's body names;
  * a multi-load asm statement whose destination overlaps its own address register is reported (a missing
    early-clobber: the first load may land before the next one is issued);
  * the kernels must not use scratch memory or spill vector registers (a 128-VGPR build that spilled faulted on the
    GPU in round 2: gpurun_out/timeline3.log; whatever the mechanism, a spilling evaluation kernel is not shipped).

Usage: isa_check_async.py <file.s> [--allow-scratch] [--min-kernels N]      exit status 1 = at least one finding
(or fewer than N evaluation kernels in the file; default 5: every one the library ships).
"""
import re
import sys

RE_RANGE = re.compile(r'\bs\[(\d+):(\d+)\]')
RE_SINGLE = re.compile(r'\bs(\d+)\b')
RE_LABEL = re.compile(r'^(\.LBB\d+_\d+):')


def sgprs(text):
    out = set()
    for a, b in RE_RANGE.findall(text):
        out.update(range(int(a), int(b) + 1))
    for a in RE_SINGLE.findall(text):
        out.add(int(a))
    return out


def function_bodies(t):
    """name -> list of lines, for every function of the module"""
    out = {}
    for m in re.finditer(r'\n(_Z\w+):\s*;[^\n]*\n(.*?)\n\.Lfunc_end', t, re.S):
        out[m.group(1)] = m.group(2).split('\n')
    return out


def parse(body):
    """-> list of instructions (dict) and label -> index"""
    ins, labels = [], {}
    in_asm = False
    for raw in body:
        s = raw.strip()
        if s.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if s.startswith(';;#ASMEND'):
            in_asm = False
            continue
        lm = RE_LABEL.match(s)
        if lm:
            labels[lm.group(1)] = len(ins)
            continue
        if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
            continue
        code = s.split(';')[0].strip()
        if not code:
            continue
        ins.append({'code': code, 'asm': in_asm})
    return ins, labels


def analyse(name, body, callee_regs):
    ins, labels = parse(body)
    n = len(ins)
    findings = []
    # per-instruction transfer description
    for k, it in enumerate(ins):
        code = it['code']
        it['load'] = None
        it['wait'] = False
        it['succ'] = [k + 1] if k + 1 < n else []
        lm = re.match(r's_load_dword(?:x\d+)?\s+(s\[\d+:\d+\]|s\d+),\s*(s\[\d+:\d+\])', code)
        if it['asm'] and lm:
            it['load'] = frozenset(sgprs(lm.group(1)))
            it['addr'] = frozenset(sgprs(lm.group(2)))
        if code.startswith('s_waitcnt') and 'lgkmcnt(0)' in code:
            it['wait'] = True
        bm = re.match(r's_(cbranch_\w+|branch)\s+(\.LBB\d+_\d+)', code)
        if bm:
            tgt = labels.get(bm.group(2))
            it['succ'] = ([] if bm.group(1) == 'branch' else it['succ']) + ([tgt] if tgt is not None and tgt < n else [])
        if code.startswith(('s_endpgm', 's_setpc')):
            it['succ'] = []
    # missing early-clobber: consecutive hand-issued loads from one address whose earlier destination overlaps it
    for k in range(n - 1):
        a, b = ins[k], ins[k + 1]
        if a['load'] is not None and b['load'] is not None and a['load'] & b['addr']:
            findings.append((k, a['code'], 'destination overlaps the address of the next hand-issued load'))
    # forward data-flow: IN[k] = union of OUT[pred]
    IN = [frozenset()] * n
    seen = [False] * n
    work = [0] if n else []
    seen[0:1] = [True]
    while work:
        k = work.pop()
        it = ins[k]
        st = IN[k]
        if it['load'] is not None:
            out = st | it['load']
        elif it['wait']:
            out = frozenset()
        else:
            out = st
        for s_ in it['succ']:
            new = IN[s_] | out
            if new != IN[s_] or not seen[s_]:
                IN[s_] = new
                seen[s_] = True
                work.append(s_)
    n_loads = sum(1 for it in ins if it['load'] is not None)
    for k, it in enumerate(ins):
        if it['asm'] or not IN[k]:
            continue
        code = it['code']
        if code.startswith('s_waitcnt'):
            continue
        touched = sgprs(code)
        if code.startswith('s_swappc_b64'):
            touched = touched | callee_regs
        hit = touched & IN[k]
        if hit:
            findings.append((k, code, 's%s in flight' % sorted(hit)))
    return n_loads, findings


def check_selection_handoff(name, body):
    """The fence-free selection (fot_kernels.hip tile_done / select_instance_wave): a wave leaves its tile's partial
    result with agent-coherent stores (`sc1`: write-through the XCD's L2), waits for them (`s_waitcnt vmcnt(0)`), counts
    itself with an agent-scope atomic, and the wave that completes the count reads all partials with agent-coherent loads
    (`sc1`) -- issued behind the wait that returns the atomic.  Correct only if the ISA really looks like that: checked
    here on every build instead of trusted.  Returns a list of problems."""
    ins = [l.split(';')[0].strip() for l in body]
    ins = [l for l in ins if l and not l.startswith('.') and not l.endswith(':')]
    atom = [k for k, l in enumerate(ins) if l.startswith('global_atomic_add')]
    if not atom:
        return ['no global_atomic_add (the per-instance tile counter) found']
    probs = []
    for a in atom:
        # the partial's stores: the run of global_store* right before the wait in front of the atomic
        w = max((k for k in range(a) if ins[k].startswith('s_waitcnt') and 'vmcnt(0)' in ins[k]), default=None)
        stores = [k for k in range(max(0, (w or a) - 40), w or a) if ins[k].startswith('global_store')]
        if w is None or not stores:
            probs.append('no s_waitcnt vmcnt(0) between the partial-result stores and the tile counter')
            continue
        if any(ins[k].startswith(('global_', 'buffer_', 'flat_')) for k in range(w + 1, a)):
            probs.append('a memory instruction sits between the wait for the partial-result stores and the tile counter')
        if len(stores) < 11:
            probs.append(f'only {len(stores)} stores in front of the tile counter (a TilePart is 11)')
        for k in stores[-11:]:
            if ' sc1' not in ins[k]:
                probs.append(f'partial-result store without sc1: {ins[k]}')
        # behind the atomic: its wait, then the selecting wave's loads
        wa = next((k for k in range(a + 1, len(ins)) if ins[k].startswith('s_waitcnt') and 'vmcnt(0)' in ins[k]), None)
        loads = [k for k in range(a + 1, min(len(ins), a + 400)) if ins[k].startswith('global_load') and ' sc1' in ins[k]]
        if wa is None:
            probs.append('the tile counter is never waited for')
        elif any(k < wa for k in loads):
            probs.append('an agent-coherent load of a partial result is issued before the tile counter has returned')
        if len(loads) < 11:
            probs.append(f'only {len(loads)} agent-coherent loads behind the tile counter (the selecting wave reads 11 fields per tile)')
    return probs


def metadata(t):
    out = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', t, re.S):
        body = m.group(2)
        g = lambda key: int((re.search(key + r':\s+(\d+)', body) or [0, '0'])[1])
        out[m.group(1)] = {'vgpr': g(r'\.vgpr_count'), 'sspill': g(r'\.sgpr_spill_count'), 'vspill': g(r'\.vgpr_spill_count'),
                           'scratch': g(r'\.private_segment_fixed_size')}
    return out


def main(argv):
    path = argv[1] if len(argv) > 1 and not argv[1].startswith('--') else '/tmp/isa/fot.s'
    allow_scratch = '--allow-scratch' in argv
    min_kernels = int(argv[argv.index('--min-kernels') + 1]) if '--min-kernels' in argv else 3
    t = open(path).read()
    funcs = function_bodies(t)
    meta = metadata(t)
    # SGPRs any non-kernel device function names (callees of the kernels)
    callee_regs = set()
    for fname, body in funcs.items():
        if 'k_' not in fname:
            callee_regs |= sgprs('\n'.join(l.split(';')[0] for l in body))
    bad_total = 0
    n_kernels = 0
    for fname, body in funcs.items():
        m = re.match(r'_ZN3fot\d+(k_evaluate\w*?)E', fname)
        if not m:
            continue
        n_kernels += 1
        n_loads, findings = analyse(m.group(1), body, frozenset(callee_regs))
        md = meta.get(fname, {})
        print(f"{m.group(1)}: {n_loads} hand-issued scalar loads, {len(findings)} instructions touching a destination in flight"
              f" (vgpr {md.get('vgpr')}, sgpr spills {md.get('sspill')}, vgpr spills {md.get('vspill')}, scratch {md.get('scratch')} B)")
        for k, code, why in findings[:12]:
            print(f"   instruction {k}: {code}   ({why})")
        bad_total += len(findings)
        if '--no-handoff-check' not in argv:
            for msg in check_selection_handoff(m.group(1), body):
                print(f"   {m.group(1)}: selection hand-off: {msg}")
                bad_total += 1
        if not allow_scratch and (md.get('vspill', 0) or md.get('scratch', 0)):
            print(f"   {m.group(1)} spills vector registers / uses scratch memory: not a build to ship")
            bad_total += 1
    if n_kernels < min_kernels:                                  # k_evaluate, _split, _group
        print(f"only {n_kernels} evaluation kernels found in {path}")
        bad_total += 1
    return 1 if bad_total else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
