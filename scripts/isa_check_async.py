"""Static check of the hand-issued scalar loads in the evaluation kernels (/tmp/isa/fot.s from scripts/isa.sh).

FusedSink issues `s_load_dwordx16` / `s_load_dword` in one inline-asm statement and waits for them in a later one.  The
compiler does not know the destination registers are still in flight in between: if it spills, copies or reuses them
there (it does under scalar-register pressure), the data lands in whatever lives in those registers by then -- wrong
chunk data or a wild address.  This walks every evaluation kernel and reports the register-allocator
moves (v_writelane / v_readlane / s_mov of an SGPR) that touch a destination register of a hand-issued load before the
next wait in straight-line code (up to the next label or branch: the chunk loop's own control flow reuses the buffers
legitimately once they have been waited for).  Found with exactly this failure: a variant of k_evaluate_group with 70
spilled SGPRs parked the warm-up loads' destination in a VGPR lane while they were in flight, and faulted."""
import re
import sys

t = open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/isa/fot.s').read()
bad_total = 0
for m in re.finditer(r'\n(_ZN3fot\d+(k_evaluate\w*?)E[^\n]*):\s*;[^\n]*\n(.*?)\n\.Lfunc_end', t, re.S):
    name, body = m.group(2), m.group(3).split('\n')
    in_asm, flight, bad, n_loads = False, [], [], 0          # flight: list of (lo, hi) SGPR ranges
    for i, l in enumerate(body):
        s = l.strip()
        if s.startswith(';;#ASMSTART'):
            in_asm = True; continue
        if s.startswith(';;#ASMEND'):
            in_asm = False; continue
        if not s or s.startswith(';'):
            continue
        if s.endswith(':') or re.match(r'\.LBB\d+_\d+:', s):           # a label: other paths join -- stop tracking the chunk
            flight = [r for r in flight if r[0] == r[1]]                 # buffers; the warm-up loads' single register stays
            continue                                                     # reserved over the whole sample arithmetic
        code = s.split(';')[0]
        if in_asm:
            lm = re.match(r's_load_dword(x\d+)?\s+s(\[(\d+):(\d+)\]|(\d+))', code)
            if lm:
                lo = int(lm.group(3) or lm.group(5)); hi = int(lm.group(4) or lm.group(5))
                flight.append((lo, hi)); n_loads += 1
                continue
            if code.startswith('s_waitcnt') and 'lgkmcnt(0)' in code:
                flight = []
            continue
        if code.startswith('s_waitcnt') and 'lgkmcnt(0)' in code:   # a compiler-placed full wait also lands them
            flight = []
            continue
        if code.startswith(('s_cbranch', 's_branch', 's_setpc', 's_swappc', 's_endpgm')):
            flight = [r for r in flight if r[0] == r[1]] if not code.startswith('s_endpgm') else []
            continue
        if not flight or not code.startswith(('v_writelane', 'v_readlane', 's_mov_b32', 's_mov_b64')):
            continue
        used = set()
        for a, b in re.findall(r'\bs\[(\d+):(\d+)\]', code):
            used.update(range(int(a), int(b) + 1))
        for a in re.findall(r'\bs(\d+)\b', code):
            used.add(int(a))
        for lo, hi in flight:
            if any(lo <= r <= hi for r in used):
                bad.append((i, code.strip(), (lo, hi)))
                break
    print(f"{name}: {n_loads} hand-issued scalar loads, {len(bad)} instructions touching a destination in flight")
    for i, code, rng in bad[:12]:
        print(f"   line {i}: {code}   (s[{rng[0]}:{rng[1]}] in flight)")
    bad_total += len(bad)
sys.exit(1 if bad_total else 0)
