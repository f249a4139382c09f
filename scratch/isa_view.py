"""scratch/isa_view.py <asm file> <kernel-name substring> [from] [to]: memory / wait / matrix instructions of a kernel's ISA."""
import re, sys
s = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r'^(_Z\w+):', s, re.M) if sys.argv[2] in m.group(1)]
i = s.index(names[0] + ':'); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0; hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(body)
pat = r'scratch_|s_waitcnt|s_barrier|global_load|s_cbranch|^\.LBB|s_load|ds_read|ds_write|v_mfma|global_store|buffer_'
prev = None; rep = 0
for k, l in enumerate(body):
    if lo <= k < hi and re.search(pat, l):
        t = l.strip().split()[0]
        if t == prev and t.startswith(('v_mfma', 'global_store')): rep += 1; continue
        if rep: print(f"      ... x{rep} more"); rep = 0
        prev = t; print(k, l.strip()[:80])
